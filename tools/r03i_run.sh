#!/bin/bash
# round 3, ninth GPU call: workgroup sort for giant jobs (window tests), two mini-batches in flight (wm_map_reads_slot), config 3 again
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/${TAG:-r03i}
mkdir -p $OUT
export WM_BENCH_DISTINCT_BATCHES=2 WM_BENCH_CPU_SAMPLE=0
echo "== window + e2e + binding GPU tests =="
timeout 900 python -m pytest tests/test_window_gpu.py tests/test_e2e_gpu.py tests/test_binding_gpu.py -m gpu -q > $OUT/gputest.txt 2>&1; echo "rc=$? $SECONDS s"; tail -5 $OUT/gputest.txt
run_bench() { # tag, args..., env via WM_ENV
  local tag=$1; shift
  local t0=$SECONDS
  ( env ${WM_ENV:-WM_X=1} timeout 900 python bench.py "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $((SECONDS-t0))s $(tail -1 $OUT/bench_$tag.log | cut -c1-200)"
}
WM_ENV="WM_BENCH_CPU_SAMPLE=4096 WM_BENCH_CPU_THREADS=16" run_bench pipe2 --steps 4 --warmup 2
WM_ENV="WM_BENCH_PIPELINE=0" run_bench pipe1 --steps 4 --warmup 2
WM_ENV="WM_INFLIGHT=8192" run_bench pipe2_inflight8k --steps 4 --warmup 2
WM_ENV="WM_TRACE=1 WM_BENCH_CPU_SAMPLE=2048 WM_BENCH_CPU_THREADS=16" run_bench config3 --config 3 --steps 2 --warmup 2 --reads-per-step 8192
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
import json, glob
for f in sorted(glob.glob(os.path.join(out, "bench_*.json"))):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        h = d["host"]
        print("%-28s %.4f Gbp/s ms/step %5.0f cpu/step %.1f util %.2f | calls w %d k %d | wall idle %.0f batched w %.1f k %.1f | parity %s | cpu %s" % (
              os.path.basename(f), d["value"], d["ms_per_step"], h["process_cpu_s"] / d["steps"], h["cpu_utilisation"], h["batched_calls"]["window"], h["batched_calls"]["ksw"],
              h["idle_wall_s"], h["batched_wall_s"]["window"], h["batched_wall_s"]["ksw"], (d.get("parity") or {}).get("mismatches"), (d.get("cpu_baseline") or {}).get("value")))
    except Exception as e:
        print(f, "unreadable:", e)
PY
