#!/bin/bash
# round 3, fourth GPU call: wave priorities (s_setprio) + workgroup-scope fences A/B, contexts sweep
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/${TAG:-r03d}
mkdir -p $OUT
export WM_BENCH_DISTINCT_BATCHES=2 WM_BENCH_CPU_SAMPLE=0
echo "== GPU tests touched by the fence change =="
timeout 900 python -m pytest tests/test_window_gpu.py tests/test_aux_gpu.py tests/test_e2e_gpu.py tests/test_ksw_gpu.py -m gpu -q > $OUT/gputest.txt 2>&1; echo "rc=$? $SECONDS s"; tail -5 $OUT/gputest.txt
run_bench() { # tag, env...
  local tag=$1; shift
  local t0=$SECONDS
  ( env "$@" timeout 600 python bench.py --steps ${STEPS:-2} --warmup 1 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $((SECONDS-t0))s"
}
run_bench prio WM_X=1
run_bench noprio WM_LIBWMGPU=$ROOT/winnowmap_amd/libwmgpu_noprio.so
run_bench prio_ctx8 WM_CONTEXTS=8
run_bench prio_ctx10 WM_CONTEXTS=10
run_bench prio_ctx12 WM_CONTEXTS=12
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ["OUT"], "bench_*.json"))):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        h = d["host"]
        print("%-26s %.4f Gbp/s  ms/step %5.0f  cpu/step %.1f util %.2f  calls window %d ksw %d  wall window %.1f ksw %.1f" % (os.path.basename(f), d["value"], d["ms_per_step"], h["process_cpu_s"] / d["steps"], h["cpu_utilisation"],
              h["batched_calls"]["window"], h["batched_calls"]["ksw"], h["batched_wall_s"]["window"], h["batched_wall_s"]["ksw"]))
    except Exception as e:
        print(f, "unreadable:", e)
PY
