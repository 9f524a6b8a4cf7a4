#!/bin/bash
# Second GPU call of round 4: the stripe kernels after the LDS address-space fix. Tight timeouts everywhere: a hang must cost seconds, not minutes.
#   gpurun --timeout 780 -- 'bash tools/r04b_run.sh'
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/r04b
mkdir -p $OUT/dump
echo "== 1. stripe parity + the test that hung =="
timeout 200 python -m pytest tests/test_ksw_gpu.py -m gpu -x -q -k "stripe" > $OUT/gputest_stripe.txt 2>&1; rc=$?; echo "rc=$rc $SECONDS s"; tail -3 $OUT/gputest_stripe.txt
if [ $rc -ne 0 ]; then echo "stripe parity failed: stopping"; exit 1; fi
WM_KSW_DUMP=$OUT/dump timeout 150 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -k "workload_scale" > $OUT/gputest_workload.txt 2>&1; rc=$?; echo "workload rc=$rc $SECONDS s"; tail -3 $OUT/gputest_workload.txt
ls -la $OUT/dump | head
if [ $rc -ne 0 ]; then echo "workload test failed or hung: stopping"; exit 1; fi
echo "== 2. isolated probe: stripe classes off / on =="
WM_KSW_STRIPE=0 timeout 120 python tools/ksw_probe.py 20000 > $OUT/probe_stripe0.txt 2>&1; echo "rc=$?"; tail -8 $OUT/probe_stripe0.txt
timeout 120 python tools/ksw_probe.py 20000 > $OUT/probe_stripe1.txt 2>&1; echo "rc=$?"; tail -8 $OUT/probe_stripe1.txt
echo "== 3. bench A/B (16384 reads per step, 4 steps) =="
run_bench() { # tag, env...
  local tag=$1; shift
  local t0=$SECONDS
  ( env "$@" WM_BENCH_CPU_SAMPLE=0 timeout 150 python bench.py --steps 4 --warmup 2 --reads-per-step 16384 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $((SECONDS-t0))s $(cut -c1-100 $OUT/bench_$tag.json)"
}
run_bench stripe1
run_bench stripe0 WM_KSW_STRIPE=0
run_bench stripe1_r8_1024 WM_KSW_STRIPE_ROWS8=1024
echo "== summary ($SECONDS s) =="
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ["OUT"], "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-28s %.4f %s  ms/step %.0f cpu_s %.1f" % (os.path.basename(f), d["value"], d["unit"], d["ms_per_step"], d["host"]["process_cpu_s"]))
        for k, v in sorted(d["roofline"]["classes"].items(), key=lambda kv: -kv[1]["ms"]):
            print("    %-44s ms %8.0f cells %.3e launches %5d GCUPS %.1f" % (k, v["ms"], v["cells"], v["launches"], v["cells"] / max(v["ms"], 1e-9) / 1e6))
    except Exception as e:
        print(f, "unreadable:", e)
PY
