#!/bin/bash
# GPU call r04k: the z-drop scan of the gap fills on the device (ksw_zdwalk_kernel): unit test against the host's compile of the same walk, end-to-end
# parity, and the bench with / without it (WM_ZDWALK_HOST=1 keeps the walk on the host) for the host CPU seconds.
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/r04k
mkdir -p $OUT
timeout 200 python -m pytest tests/test_ksw_gpu.py -m gpu -x -q -k "position_jobs" > $OUT/gputest_ksw.txt 2>&1; echo "ksw rc=$? $SECONDS s"; tail -3 $OUT/gputest_ksw.txt
timeout 200 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q > $OUT/gputest_e2e.txt 2>&1; echo "e2e rc=$? $SECONDS s"; tail -3 $OUT/gputest_e2e.txt
timeout 200 python -m pytest tests/test_binding_gpu.py -m gpu -x -q -k "bound_to_the_library or below_the_mcas" > $OUT/gputest_binding.txt 2>&1; echo "binding rc=$? $SECONDS s"; tail -3 $OUT/gputest_binding.txt
run_bench() { # tag, env...
  local tag=$1; shift
  local t0=$SECONDS
  ( env "$@" WM_BENCH_FILE=0 WM_BENCH_CPU_SAMPLE=0 timeout 150 python bench.py --steps 8 --warmup 4 --reads-per-step 16384 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $((SECONDS-t0))s $(cut -c1-75 $OUT/bench_$tag.json | cut -c30-75)"
}
run_bench dev_a
run_bench host_a WM_ZDWALK_HOST=1
run_bench dev_b
run_bench host_b WM_ZDWALK_HOST=1
echo "== summary ($SECONDS s) =="
python - <<'PY'
import json, glob, os
out = os.environ["OUT"]
for f in sorted(glob.glob(os.path.join(out, "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        h = d["host"]
        print("%-12s %.4f Gbp/s ms/step %5.0f cpu_s %.1f glue_cpu %.1f util %.2f host_bound %.3f hits %d" % (os.path.basename(f)[6:-5], d["value"], d["ms_per_step"], h["process_cpu_s"], h["glue_cpu_s"], h["cpu_utilisation"], h["host_bound_gbps_per_rank"], d["config"]["hits"]))
    except Exception as e:
        print(f, "unreadable:", e)
PY
