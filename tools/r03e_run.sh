#!/bin/bash
# round 3, fifth GPU call: lazy / vector-unit exact maximum in the DP kernels (probe + ksw tests), reads-in-flight and context sweeps
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/${TAG:-r03e}
mkdir -p $OUT
export WM_BENCH_DISTINCT_BATCHES=2 WM_BENCH_CPU_SAMPLE=0
echo "== ksw + e2e + binding GPU tests =="
timeout 900 python -m pytest tests/test_ksw_gpu.py tests/test_e2e_gpu.py tests/test_binding_gpu.py -m gpu -q > $OUT/gputest.txt 2>&1; echo "rc=$? $SECONDS s"; tail -5 $OUT/gputest.txt
echo "== probe =="
timeout 400 python tools/ksw_probe.py > $OUT/ksw_probe.txt 2>&1; cat $OUT/ksw_probe.txt | tail -9
run_bench() { # tag, env...
  local tag=$1; shift
  local t0=$SECONDS
  ( env "$@" timeout 600 python bench.py --steps ${STEPS:-2} --warmup 1 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $((SECONDS-t0))s"
}
run_bench default WM_X=1
run_bench inflight32k WM_INFLIGHT=32768
run_bench inflight64k WM_INFLIGHT=65536
run_bench ctx5 WM_CONTEXTS=5
run_bench ctx4_inflight32k WM_CONTEXTS=4 WM_INFLIGHT=32768
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ["OUT"], "bench_*.json"))):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        h = d["host"]
        print("%-30s %.4f Gbp/s  ms/step %5.0f  cpu/step %.1f util %.2f  calls window %d ksw %d  wall window %.1f ksw %.1f" % (os.path.basename(f), d["value"], d["ms_per_step"], h["process_cpu_s"] / d["steps"], h["cpu_utilisation"],
              h["batched_calls"]["window"], h["batched_calls"]["ksw"], h["batched_wall_s"]["window"], h["batched_wall_s"]["ksw"]))
    except Exception as e:
        print(f, "unreadable:", e)
d = json.loads([l for l in open(os.path.join(os.environ["OUT"], "bench_default.json")).read().splitlines() if l.startswith("{")][-1])
for k, v in sorted(d["roofline"]["classes"].items(), key=lambda kv: -kv[1]["ms"]):
    print("  %-44s ms %9.0f cells %.3e launches %5d  GCUPS %.1f" % (k, v["ms"], v["cells"], v["launches"], v["cells"] / max(v["ms"], 1e-9) / 1e6))
PY
