#!/bin/bash
# round 3, tenth GPU call: (1) isolated ksw probe with three builds — default (loads_land: no per-row store drain in the register kernels), the
# round-2 code (WM_LOADS_LAND=0) and a timing-only no-store build; (2) the full GPU suite (splice mode, 1-Gbase capacity, window kernels …);
# (3) bench A/B (new / old build); (4) HBM-traffic counters + kernel stats of the final build (tools/prof_bench.sh, PMC=1)
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/${TAG:-r03j}
mkdir -p $OUT
echo "== probe: default / round-2 wait placement / no stores =="
for v in "" _ll0 _nostore; do
  ( WM_LIBWMGPU=$ROOT/winnowmap_amd/libwmgpu$v.so timeout 240 python tools/ksw_probe.py 20000 > $OUT/ksw_probe$v.txt 2>&1 ); echo "probe$v rc=$?"; head -8 $OUT/ksw_probe$v.txt
done
run_bench() { # tag, args..., env via WM_ENV
  local tag=$1; shift
  local t0=$SECONDS
  ( env ${WM_ENV:-WM_X=1} timeout 900 python bench.py "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $((SECONDS-t0))s $(tail -1 $OUT/bench_$tag.log | cut -c1-200)"
}
export WM_BENCH_DISTINCT_BATCHES=2
echo "== bench with the round-2 wait placement (the default build runs under prof_bench below) =="
WM_ENV="WM_BENCH_CPU_SAMPLE=0 WM_LIBWMGPU=$ROOT/winnowmap_amd/libwmgpu_ll0.so" run_bench ll0 --steps 4 --warmup 2
echo "== full GPU suite =="
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/gputest.txt 2>&1; echo "rc=$? $SECONDS s"; tail -8 $OUT/gputest.txt
echo "== counters + stats of the final build =="
unset WM_BENCH_DISTINCT_BATCHES
PMC=1 PMC_READS=4096 timeout 1200 bash tools/prof_bench.sh r03j_bench > $OUT/prof_bench.txt 2>&1; echo "prof rc=$? $SECONDS s"; tail -40 $OUT/prof_bench.txt
python - <<'PY'
import os, json, glob
out = os.environ["OUT"]
for f in sorted(glob.glob(os.path.join(out, "bench_*.json"))):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        h = d["host"]
        print("%-28s %.4f Gbp/s ms/step %5.0f cpu/step %.1f util %.2f | calls w %d k %d | parity %s | cpu %s | defines [%s]" % (
              os.path.basename(f), d["value"], d["ms_per_step"], h["process_cpu_s"] / d["steps"], h["cpu_utilisation"], h["batched_calls"]["window"], h["batched_calls"]["ksw"],
              (d.get("parity") or {}).get("mismatches"), (d.get("cpu_baseline") or {}).get("value"), d["roofline"].get("variants", {}).get("kernel_defines")))
        for k, v in sorted(d["roofline"]["classes"].items(), key=lambda kv: -kv[1]["ms"])[:8]:
            print("    %-44s ms %8.0f cells %.3e launches %5d GCUPS %.1f" % (k, v["ms"], v["cells"], v["launches"], v["cells"] / max(v["ms"], 1e-9) / 1e6))
    except Exception as e:
        print(f, "unreadable:", e)
PY
echo done
