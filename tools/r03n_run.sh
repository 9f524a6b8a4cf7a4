#!/bin/bash
# round 3, GPU call N: A/B of the cheaper approximate-max track for unclipped fills (WM_KSW_EDGE_TRACK=1 default vs 0): isolated probe + bench
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/${TAG:-r03n}
mkdir -p $OUT
for v in "" _et0; do
  ( WM_LIBWMGPU=$ROOT/winnowmap_amd/libwmgpu$v.so timeout 240 python tools/ksw_probe.py 20000 > $OUT/ksw_probe$v.txt 2>&1 ); echo "probe$v rc=$?"; head -9 $OUT/ksw_probe$v.txt
done
echo "== ksw GPU tests with the default build =="
timeout 600 python -m pytest tests/test_ksw_gpu.py -m gpu -q -x > $OUT/gputest.txt 2>&1; echo "rc=$? $SECONDS s"; tail -3 $OUT/gputest.txt
run_bench() { local tag=$1; shift; local t0=$SECONDS
  ( env ${WM_ENV:-WM_X=1} timeout 900 python bench.py "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $((SECONDS-t0))s $(tail -1 $OUT/bench_$tag.log | cut -c1-200)"; }
export WM_BENCH_DISTINCT_BATCHES=2 WM_BENCH_CPU_SAMPLE=0
WM_ENV="WM_BENCH_CPU_SAMPLE=4096 WM_BENCH_CPU_THREADS=16" run_bench new --steps 4 --warmup 2
WM_ENV="WM_LIBWMGPU=$ROOT/winnowmap_amd/libwmgpu_et0.so" run_bench et0 --steps 4 --warmup 2
python - <<'PY'
import os, json, glob
out = os.environ["OUT"]
for f in sorted(glob.glob(os.path.join(out, "bench_*.json"))):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        h = d["host"]
        print("%-22s %.4f Gbp/s ms/step %5.0f cpu/step %.1f | defines [%s]" % (os.path.basename(f), d["value"], d["ms_per_step"], h["process_cpu_s"] / d["steps"], d["config"].get("variants", {}).get("kernel_defines")))
        for k, v in sorted(d["roofline"]["classes"].items(), key=lambda kv: -kv[1]["ms"])[:8]:
            print("    %-44s ms %8.0f cells %.3e launches %5d GCUPS %.1f" % (k, v["ms"], v["cells"], v["launches"], v["cells"] / max(v["ms"], 1e-9) / 1e6))
    except Exception as e:
        print(f, "unreadable:", e)
PY
echo done
