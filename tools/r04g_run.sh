#!/bin/bash
# GPU call r04g: hub / queue knobs on top of 20 hardware queues + split side pool; stripe classes on/off; two runs each (run-to-run noise is +-4 %).
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/r04g
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=20 WM_SIDE_SPLIT=6,4
run_bench() { # tag, env...
  local tag=$1; shift
  local t0=$SECONDS
  ( env "$@" WM_BENCH_CPU_SAMPLE=0 timeout 150 python bench.py --steps 6 --warmup 2 --reads-per-step 16384 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $((SECONDS-t0))s $(cut -c1-75 $OUT/bench_$tag.json | cut -c30-75)"
}
for rep in a b; do
run_bench base_$rep WM_KSW_STRIPE=0
run_bench stripe_$rep
run_bench stripe_r8_4096_$rep WM_KSW_STRIPE_ROWS8=4096
run_bench ctx8_$rep WM_KSW_STRIPE=0 WM_CONTEXTS=8 WM_SIDE_SPLIT=5,4
run_bench xmax2_$rep WM_KSW_STRIPE=0 WM_KSWX_MAX=2 WM_KSWH_MAX=3
run_bench wait_$rep WM_KSW_STRIPE=0 WM_KSWH_MAX_WAIT_MS=30 WM_KSWX_MAX_WAIT_MS=50 WM_KSW_MAX_WAIT_MS=15
run_bench huge64k_$rep WM_KSW_STRIPE=0 WM_KSW_HUGE_UNITS=65536
done
echo "== summary ($SECONDS s) =="
python - <<'PY'
import json, glob, os
out = os.environ["OUT"]
for f in sorted(glob.glob(os.path.join(out, "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        h = d["host"]
        print("%-24s %.4f Gbp/s ms/step %5.0f cpu_s %.1f  calls %s wall %s idle %.0f" % (os.path.basename(f)[6:-5], d["value"], d["ms_per_step"], h["process_cpu_s"], h["batched_calls"], h["batched_wall_s"], h["idle_wall_s"]))
    except Exception as e:
        print(f, "unreadable:", e)
PY
