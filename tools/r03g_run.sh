#!/bin/bash
# round 3, seventh GPU call: multi-wave DP kernel without FLAT sequence loads (no store drain per row): ksw tests, bench, kernel-trace stats
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/${TAG:-r03g}
mkdir -p $OUT
export WM_BENCH_DISTINCT_BATCHES=2 WM_BENCH_CPU_SAMPLE=0
echo "== ksw GPU tests =="
timeout 900 python -m pytest tests/test_ksw_gpu.py tests/test_e2e_gpu.py -m gpu -q > $OUT/gputest.txt 2>&1; echo "rc=$? $SECONDS s"; tail -5 $OUT/gputest.txt
( timeout 600 python bench.py --steps 2 --warmup 1 > $OUT/bench_default.json 2> $OUT/bench_default.log ); echo "[default] rc=$? $SECONDS s"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 > $OUT/bench_prof.json 2> $OUT/bench_prof.log ); echo "prof rc=$? $SECONDS s"
python - <<'PY'
import os, sqlite3, glob, json
out = os.environ["OUT"]
for f in ("bench_default.json", "bench_prof.json"):
    try:
        d = json.loads([l for l in open(os.path.join(out, f)).read().splitlines() if l.startswith("{")][-1])
        h = d["host"]
        print("%-20s %.4f Gbp/s ms/step %5.0f cpu/step %.1f util %.2f | calls w %d k %d | wall idle %.0f batched w %.1f k %.1f" % (f, d["value"], d["ms_per_step"], h["process_cpu_s"] / d["steps"], h["cpu_utilisation"],
              h["batched_calls"]["window"], h["batched_calls"]["ksw"], h["idle_wall_s"], h["batched_wall_s"]["window"], h["batched_wall_s"]["ksw"]))
        if f == "bench_default.json":
            for k, v in sorted(d["roofline"]["classes"].items(), key=lambda kv: -kv[1]["ms"]):
                print("  %-44s ms %9.0f cells %.3e launches %5d  GCUPS %.1f" % (k, v["ms"], v["cells"], v["launches"], v["cells"] / max(v["ms"], 1e-9) / 1e6))
    except Exception as e:
        print(f, "unreadable:", e)
for f in glob.glob(out + "/stats/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    print("%-100s %8s %12s %10s %6s" % ("kernel", "calls", "total_s", "avg_ms", "pct"))
    for r in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 32"):
        print("%-100s %8d %12.2f %10.2f %6.2f" % (r[0][:100], r[1], r[2] / 1e9, r[3] / 1e6, r[4]))
PY
for f in $(find $OUT/stats -name "*.db"); do python tools/gpu_timeline.py $f > $OUT/timeline.txt 2>&1; done; head -8 $OUT/timeline.txt
find $OUT/stats -name "*.db" -size +30M -delete
