#!/bin/bash
# GPU call r04m: the host's CPU goes into mprotect (r04l: 58 % of the samples) = glibc malloc growing / deleting the heaps of the worker threads' arenas.
# A/B with the allocator told to keep its memory (environment only), and the sampling profile of the tuned run.
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/r04m
mkdir -p $OUT
g++ -O2 -fPIC -shared -o /tmp/libsprof.so tools/sprof/sprof.cpp -ldl
run_bench() { # tag, env...
  local tag=$1; shift
  local t0=$SECONDS
  ( env "$@" WM_BENCH_FILE=0 WM_BENCH_CPU_SAMPLE=0 timeout 150 python bench.py --steps 8 --warmup 4 --reads-per-step 16384 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $((SECONDS-t0))s $(cut -c1-75 $OUT/bench_$tag.json | cut -c30-75)"
}
TUNE="MALLOC_TOP_PAD_=67108864 MALLOC_TRIM_THRESHOLD_=4294967296 MALLOC_MMAP_THRESHOLD_=33554432"
run_bench base_a
run_bench tuned_a $TUNE
run_bench tuned_b $TUNE
run_bench arena4 $TUNE MALLOC_ARENA_MAX=4
run_bench tuned_sprof $TUNE SPROF_OUT=$OUT/sprof.txt LD_PRELOAD=/tmp/libsprof.so
echo "== summary ($SECONDS s) =="
python - <<'PY'
import json, glob, os
out = os.environ["OUT"]
for f in sorted(glob.glob(os.path.join(out, "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        h = d["host"]
        print("%-12s %.4f Gbp/s ms/step %5.0f cpu_s %.1f glue_cpu %.1f util %.2f host_bound %.3f wall %s hits %d" % (os.path.basename(f)[6:-5], d["value"], d["ms_per_step"], h["process_cpu_s"], h["glue_cpu_s"], h["cpu_utilisation"], h["host_bound_gbps_per_rank"], h["batched_wall_s"], d["config"]["hits"]))
    except Exception as e:
        print(f, "unreadable:", e)
PY
for f in $OUT/sprof.txt.*; do python tools/sprof/resolve.py $f 60 > $f.resolved 2>&1; head -40 $f.resolved; done
