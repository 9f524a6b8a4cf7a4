#!/bin/bash
# round 3, sixth GPU call: fused small-job window kernel + direct result pools; hub wall-time accounting; probe
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/${TAG:-r03f}
mkdir -p $OUT
export WM_BENCH_DISTINCT_BATCHES=2 WM_BENCH_CPU_SAMPLE=0
echo "== window + aux + e2e GPU tests =="
timeout 900 python -m pytest tests/test_window_gpu.py tests/test_aux_gpu.py tests/test_e2e_gpu.py -m gpu -q > $OUT/gputest.txt 2>&1; echo "rc=$? $SECONDS s"; tail -5 $OUT/gputest.txt
run_bench() { # tag, env...
  local tag=$1; shift
  local t0=$SECONDS
  ( env "$@" timeout 600 python bench.py --steps ${STEPS:-2} --warmup 1 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $((SECONDS-t0))s"
}
run_bench default WM_X=1
run_bench waits WM_KSWH_MAX_WAIT_MS=20 WM_KSWX_MAX_WAIT_MS=30 WM_KSW_MAX_WAIT_MS=15
run_bench threads12 WM_BENCH_THREADS=12
run_bench parity WM_BENCH_CPU_SAMPLE=8192 WM_BENCH_CPU_THREADS=16
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ["OUT"], "bench_*.json"))):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        h = d["host"]
        print("%-24s %.4f Gbp/s ms/step %5.0f cpu/step %.1f util %.2f | calls w %d k %d | wall: workers %.0f glue %.0f (cpu %.0f) lock %.1f idle %.0f batched w %.1f k %.1f | throttled %s | parity %s" % (
              os.path.basename(f), d["value"], d["ms_per_step"], h["process_cpu_s"] / d["steps"], h["cpu_utilisation"], h["batched_calls"]["window"], h["batched_calls"]["ksw"],
              h["workers_wall_s"], h["glue_wall_s"], h["glue_cpu_s"], h["lock_wait_wall_s"], h["idle_wall_s"], h["batched_wall_s"]["window"], h["batched_wall_s"]["ksw"], h.get("cpu_quota_throttled"),
              (d.get("parity") or {}).get("mismatches")))
    except Exception as e:
        print(f, "unreadable:", e)
PY
