#!/bin/bash
# round 3, third GPU call: where does a window call's time go? kernel-trace stats + the hub's per-batch trace; context-count sweep
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/${TAG:-r03c}
mkdir -p $OUT
export WM_BENCH_DISTINCT_BATCHES=2 WM_BENCH_CPU_SAMPLE=0
echo "== traced bench under rocprofv3 --kernel-trace --stats =="
( cd /tmp && WM_TRACE=1 timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 > $OUT/bench_prof.json 2> $OUT/bench_prof.log ); echo "rc=$? $SECONDS s"
grep -c "^\[batch\]" $OUT/bench_prof.log
python - <<'PY'
import os, re, collections, sqlite3, glob, json
out = os.environ["OUT"]
agg = collections.defaultdict(list)
for l in open(out + "/bench_prof.log", errors="ignore"):
    m = re.match(r"\[batch\] worker\s+\d+ (\S+) n=(\d+) ([0-9.]+) ms", l)
    if m: agg[m.group(1)].append((int(m.group(2)), float(m.group(3))))
for k, v in agg.items():
    ms = sorted(x[1] for x in v); ns = [x[0] for x in v]
    print("%-10s calls %5d  reqs/call avg %8.0f  ms: avg %7.1f p50 %7.1f p90 %7.1f max %8.1f" % (k, len(v), sum(ns) / len(ns), sum(ms) / len(ms), ms[len(ms) // 2], ms[int(len(ms) * .9)], ms[-1]))
for l in open(out + "/bench_prof.log", errors="ignore"):
    if l.startswith("[site]") or l.startswith("[ops") or l.startswith("[host]"): print(l.rstrip()[:300])
for f in glob.glob(out + "/stats/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    print("%-100s %8s %12s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
    for r in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 45"):
        print("%-100s %8d %12.1f %10.0f %6.2f" % (r[0][:100], r[1], r[2] / 1e6, r[3] / 1e3, r[4]))
try:
    d = json.loads([l for l in open(out + "/bench_prof.json").read().splitlines() if l.startswith("{")][-1])
    print("traced+profiled value %.4f ms/step %.0f host %s" % (d["value"], d["ms_per_step"], json.dumps(d["host"])))
except Exception as e:
    print("bench json:", e)
PY
for f in $(find $OUT/stats -name "*.db"); do python tools/gpu_timeline.py $f > $OUT/timeline.txt 2>&1; done; cat $OUT/timeline.txt | head -40
find $OUT/stats -name "*.db" -size +30M -delete
echo "== contexts sweep =="
for C in 10 14; do
  ( WM_CONTEXTS=$C timeout 600 python bench.py --steps 2 --warmup 1 > $OUT/bench_ctx$C.json 2> $OUT/bench_ctx$C.log ); echo "[ctx $C] rc=$? $SECONDS s $(cut -c1-160 $OUT/bench_ctx$C.json)"
done
