#!/bin/bash
# round 3, last GPU call: the two opt-in switches prepared for round 4 (WM_KSW_LAZY_RED, WM_KSW_PMULTI_EDGE; default build unchanged): probe, ksw GPU tests, bench A/B
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/${TAG:-r03t}
mkdir -p $OUT
for v in "" _next; do
  ( WM_LIBWMGPU=$ROOT/winnowmap_amd/libwmgpu$v.so timeout 200 python tools/ksw_probe.py 20000 > $OUT/ksw_probe$v.txt 2>&1 ); echo "probe$v rc=$?"; head -12 $OUT/ksw_probe$v.txt
done
( WM_LIBWMGPU=$ROOT/winnowmap_amd/libwmgpu_next.so timeout 300 python -m pytest tests/test_ksw_gpu.py -m gpu -q -x > $OUT/gputest_next.txt 2>&1 ); echo "ksw tests (next) rc=$?"; tail -2 $OUT/gputest_next.txt
run_bench() { local tag=$1; shift; local t0=$SECONDS
  ( env ${WM_ENV:-WM_X=1} timeout 600 python bench.py "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $((SECONDS-t0))s $(tail -1 $OUT/bench_$tag.log | cut -c1-160)"; }
export WM_BENCH_DISTINCT_BATCHES=2
WM_ENV="WM_LIBWMGPU=$ROOT/winnowmap_amd/libwmgpu_next.so WM_BENCH_CPU_SAMPLE=2048 WM_BENCH_CPU_THREADS=16" run_bench next --steps 4 --warmup 2
WM_ENV="WM_BENCH_CPU_SAMPLE=0" run_bench default --steps 4 --warmup 2
python - <<'PY'
import os, json, glob
out = os.environ["OUT"]
for f in sorted(glob.glob(os.path.join(out, "bench_*.json"))):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print("%-22s %.4f Gbp/s ms/step %5.0f | parity %s | defines [%s]" % (os.path.basename(f), d["value"], d["ms_per_step"], (d.get("parity") or {}).get("mismatches"), d["config"].get("variants", {}).get("kernel_defines")))
        for k, v in sorted(d["roofline"]["classes"].items(), key=lambda kv: -kv[1]["ms"])[:8]:
            print("    %-44s ms %8.0f cells %.3e launches %5d GCUPS %.1f" % (k, v["ms"], v["cells"], v["launches"], v["cells"] / max(v["ms"], 1e-9) / 1e6))
    except Exception as e:
        print(f, "unreadable:", e)
PY
echo done
