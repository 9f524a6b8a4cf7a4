#!/bin/bash
# GPU call r04d: where does the time of a batched call go? one short traced bench + a sweep of scheduling knobs (each run ~30 s).
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/r04d
mkdir -p $OUT
run_bench() { # tag, env...
  local tag=$1; shift
  local t0=$SECONDS
  ( env "$@" WM_BENCH_CPU_SAMPLE=0 timeout 150 python bench.py --steps 4 --warmup 2 --reads-per-step 16384 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $((SECONDS-t0))s $(cut -c1-75 $OUT/bench_$tag.json | cut -c30-75)"
}
run_bench trace WM_TRACE=1 WM_KSW_STRIPE=0
grep -c "\[batch\]" $OUT/bench_trace.log
run_bench base WM_KSW_STRIPE=0
run_bench ctx8 WM_KSW_STRIPE=0 WM_CONTEXTS=8
run_bench ctx12 WM_KSW_STRIPE=0 WM_CONTEXTS=12 GPU_MAX_HW_QUEUES=24
run_bench kswx2 WM_KSW_STRIPE=0 WM_KSWX_MAX=2 WM_KSWH_MAX=3
run_bench ctx10_q32 WM_KSW_STRIPE=0 WM_CONTEXTS=10 GPU_MAX_HW_QUEUES=32 WM_KSWX_MAX=2 WM_KSWH_MAX=3 WM_KSW_MAX=4 WM_WINDOW_MAX=4
run_bench stripe_ctx8 WM_CONTEXTS=8
run_bench inflight32k WM_KSW_STRIPE=0 WM_INFLIGHT=32768
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
# traced run: per operation, the distribution of batched-call durations
ops = collections.defaultdict(list)
for l in open(os.path.join(out, "bench_trace.log"), errors="replace"):
    m = re.match(r"\[batch\] worker\s+\d+ (\S+) n=(\d+) ([0-9.]+) ms", l)
    if m:
        ops[m.group(1)].append((float(m.group(3)), int(m.group(2))))
for op, v in sorted(ops.items()):
    ms = sorted(x[0] for x in v); n = sum(x[1] for x in v)
    print("%-10s calls %5d reqs %9d  ms: mean %.1f p50 %.1f p90 %.1f max %.1f  sum %.1f s" % (op, len(v), n, sum(ms) / len(ms), ms[len(ms) // 2], ms[int(len(ms) * 0.9)], ms[-1], sum(ms) / 1e3))
PY
grep "\[ops, sum" $OUT/bench_trace.log | tail -4
