#!/bin/bash
# round 3, eighth GPU call: wide-hull kernels alone on the chip (probe), BASELINE config 3 (HiFi) on the fused window path, hub policy sweep
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/${TAG:-r03h}
mkdir -p $OUT
export WM_BENCH_DISTINCT_BATCHES=2 WM_BENCH_CPU_SAMPLE=0
echo "== probe =="
timeout 600 python tools/ksw_probe.py > $OUT/ksw_probe.txt 2>&1; tail -12 $OUT/ksw_probe.txt
run_bench() { # tag, args..., env via WM_ENV
  local tag=$1; shift
  local t0=$SECONDS
  ( env ${WM_ENV:-WM_X=1} timeout 900 python bench.py "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $((SECONDS-t0))s $(tail -1 $OUT/bench_$tag.log | cut -c1-200)"
}
echo "== config 3 (HiFi), 8192 reads per step, traced =="
WM_ENV="WM_TRACE=1 WM_BENCH_CPU_SAMPLE=2048 WM_BENCH_CPU_THREADS=16" run_bench config3 --config 3 --steps 1 --warmup 1 --reads-per-step 8192
python - <<'PY'
import os, re, collections
out = os.environ["OUT"]
agg = collections.defaultdict(list)
for l in open(out + "/bench_config3.log", errors="ignore"):
    m = re.match(r"\[batch\] worker\s+\d+ (\S+) n=(\d+) ([0-9.]+) ms", l)
    if m: agg[m.group(1)].append((int(m.group(2)), float(m.group(3))))
for k, v in agg.items():
    ms = sorted(x[1] for x in v); ns = [x[0] for x in v]
    print("config3 %-10s calls %5d  reqs/call avg %8.0f  ms: avg %7.1f p50 %7.1f p90 %7.1f max %8.1f" % (k, len(v), sum(ns) / len(ns), sum(ms) / len(ms), ms[len(ms) // 2], ms[int(len(ms) * .9)], ms[-1]))
PY
echo "== hub policy sweep (config 2) =="
WM_ENV="WM_KSWH_MAX=3 WM_KSWX_MAX=2 WM_WINDOW_MAX=2" run_bench heavy3 --steps 2 --warmup 1
WM_ENV="WM_KSW_HEAVY_UNITS=16384 WM_KSW_HUGE_UNITS=262144" run_bench units2x --steps 2 --warmup 1
WM_ENV="WM_KSW_HEAVY_UNITS=4096 WM_KSW_HUGE_UNITS=65536" run_bench unitshalf --steps 2 --warmup 1
WM_ENV="WM_SIDE_STREAMS=4 WM_CONTEXTS=5" run_bench side4ctx5 --steps 2 --warmup 1
WM_ENV="WM_KSW_MIN_BATCH=49152 WM_WINDOW_MIN_BATCH=4096 WM_KSWH_MIN_BATCH=6144" run_bench halfbatch --steps 2 --warmup 1
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ["OUT"], "bench_*.json"))):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        h = d["host"]
        print("%-24s %.4f Gbp/s ms/step %5.0f cpu/step %.1f util %.2f | calls w %d k %d | wall idle %.0f batched w %.1f k %.1f | parity %s | cpu %s" % (
              os.path.basename(f), d["value"], d["ms_per_step"], h["process_cpu_s"] / d["steps"], h["cpu_utilisation"], h["batched_calls"]["window"], h["batched_calls"]["ksw"],
              h["idle_wall_s"], h["batched_wall_s"]["window"], h["batched_wall_s"]["ksw"], (d.get("parity") or {}).get("mismatches"), (d.get("cpu_baseline") or {}).get("value")))
    except Exception as e:
        print(f, "unreadable:", e)
PY
