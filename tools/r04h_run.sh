#!/bin/bash
# GPU call r04h: second window queue (stage-2 / long windows apart from the 1-kb stage-1 windows): parity + bench A/B, two runs each; traced run.
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/r04h
mkdir -p $OUT
timeout 200 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q > $OUT/gputest.txt 2>&1; echo "e2e rc=$?"; tail -2 $OUT/gputest.txt
run_bench() { # tag, env...
  local tag=$1; shift
  local t0=$SECONDS
  ( env "$@" WM_BENCH_FILE=0 WM_BENCH_CPU_SAMPLE=0 timeout 150 python bench.py --steps 6 --warmup 2 --reads-per-step 16384 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $((SECONDS-t0))s $(cut -c1-75 $OUT/bench_$tag.json | cut -c30-75)"
}
for rep in a b; do
run_bench twoq_$rep WM_KSW_STRIPE=0
run_bench oneq_$rep WM_KSW_STRIPE=0 WM_WINDOW_BIG_LEN=-1
run_bench twoq_stripe_$rep
done
run_bench twoq_trace WM_KSW_STRIPE=0 WM_TRACE=1
run_bench twoq_w3 WM_KSW_STRIPE=0 WM_WINDOW_MAX_WAIT_MS=5 WM_WINDOW_MIN_BATCH=4096
echo "== summary ($SECONDS s) =="
python - <<'PY'
import json, glob, os, re, collections
out = os.environ["OUT"]
for f in sorted(glob.glob(os.path.join(out, "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        h = d["host"]
        print("%-24s %.4f Gbp/s ms/step %5.0f cpu_s %.1f  calls %s wall %s idle %.0f" % (os.path.basename(f)[6:-5], d["value"], d["ms_per_step"], h["process_cpu_s"], h["batched_calls"], h["batched_wall_s"], h["idle_wall_s"]))
    except Exception as e:
        print(f, "unreadable:", e)
ops = collections.defaultdict(list)
for l in open(os.path.join(out, "bench_twoq_trace.log"), errors="replace"):
    m = re.match(r"\[batch\] worker\s+\d+ (\S+) n=(\d+) ([0-9.]+) ms", l)
    if m:
        ops[m.group(1)].append((float(m.group(3)), int(m.group(2))))
for op, v in sorted(ops.items()):
    ms = sorted(x[0] for x in v); n = sum(x[1] for x in v)
    print("%-10s calls %5d reqs %9d  ms: mean %.1f p50 %.1f p90 %.1f max %.1f  sum %.1f s" % (op, len(v), n, sum(ms) / len(ms), ms[len(ms) // 2], ms[int(len(ms) * 0.9)], ms[-1], sum(ms) / 1e3))
PY
