#!/bin/bash
# GPU call r04f: (1) kernel trace of a short bench, dumped as CSV for offline timeline analysis; (2) hardware-queue / side-pool sweep, two runs each.
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/r04f
mkdir -p $OUT
( cd /tmp && WM_KSW_STRIPE=0 WM_BENCH_CPU_SAMPLE=0 timeout 200 rocprofv3 --kernel-trace -d $OUT/trace -o bench -- python $ROOT/bench.py --steps 2 --warmup 2 --reads-per-step 16384 > $OUT/bench_traced.json 2> $OUT/bench_traced.log ); echo "trace rc=$?"
python - <<'PY'
import sqlite3, glob, os, csv, gzip
out = os.environ["OUT"]
for f in glob.glob(os.path.join(out, "trace", "**", "*.db"), recursive=True):
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    open(os.path.join(out, "schema.txt"), "w").write("\n".join(tabs))
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    open(os.path.join(out, "schema.txt"), "a").write("\nkernels: " + ",".join(cols))
    want = [c for c in ("name", "start", "end", "queue_id", "stream_id", "grid_x", "workgroup_x", "tid") if c in cols]
    with gzip.open(os.path.join(out, "kernels.csv.gz"), "wt") as fo:
        w = csv.writer(fo); w.writerow(want)
        for r in db.execute("select %s from kernels order by start" % ",".join(want)):
            r = list(r); r[0] = r[0].split("(")[0][:60]
            w.writerow(r)
    db.close(); os.remove(f)
PY
ls -la $OUT | head; rm -rf $OUT/trace
run_bench() { # tag, env...
  local tag=$1; shift
  local t0=$SECONDS
  ( env "$@" WM_KSW_STRIPE=0 WM_BENCH_CPU_SAMPLE=0 timeout 150 python bench.py --steps 6 --warmup 2 --reads-per-step 16384 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $((SECONDS-t0))s $(cut -c1-75 $OUT/bench_$tag.json | cut -c30-75)"
}
for rep in a b; do
run_bench q16_433_$rep
run_bench q20_644_$rep GPU_MAX_HW_QUEUES=20 WM_SIDE_SPLIT=6,4
run_bench q24_855_$rep GPU_MAX_HW_QUEUES=24 WM_SIDE_SPLIT=8,5
run_bench q32_c6_$rep GPU_MAX_HW_QUEUES=32 WM_SIDE_SPLIT=12,8
done
run_bench q24_c8 GPU_MAX_HW_QUEUES=24 WM_CONTEXTS=8 WM_SIDE_SPLIT=8,4
run_bench q20_nosplit GPU_MAX_HW_QUEUES=20 WM_SIDE_SPLIT=0,0
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
