#!/bin/bash
# GPU call r04j: (1) validation of the round's host-side / index / multi-mapper work (as r04i, after the test fixes), (2) mini-batches in flight:
# bench A/B with 2 / 3 / 4 slots (WM_BENCH_SLOTS), 16 host threads each, and with more contexts for the 4-slot case.
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/r04j
mkdir -p $OUT
timeout 150 python -m pytest tests/test_aux_gpu.py -m gpu -x -q -k "index_build_on_device or refuses or kmer" > $OUT/gputest_aux.txt 2>&1; echo "aux rc=$? $SECONDS s"; tail -3 $OUT/gputest_aux.txt
timeout 200 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q > $OUT/gputest_e2e.txt 2>&1; echo "e2e rc=$? $SECONDS s"; tail -3 $OUT/gputest_e2e.txt
timeout 300 python -m pytest tests/test_binding_gpu.py -m gpu -x -q -k "below_the_mcas or split_prefix" > $OUT/gputest_binding.txt 2>&1; echo "binding rc=$? $SECONDS s"; tail -4 $OUT/gputest_binding.txt
run_bench() { # tag, env...
  local tag=$1; shift
  local t0=$SECONDS
  ( env "$@" WM_BENCH_FILE=0 WM_BENCH_CPU_SAMPLE=0 timeout 150 python bench.py --steps 8 --warmup 4 --reads-per-step 16384 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $((SECONDS-t0))s $(cut -c1-75 $OUT/bench_$tag.json | cut -c30-75)"
}
run_bench s2_a WM_BENCH_SLOTS=2
run_bench s3_a WM_BENCH_SLOTS=3
run_bench s4_a WM_BENCH_SLOTS=4
run_bench s4c8 WM_BENCH_SLOTS=4 WM_CONTEXTS=8
run_bench s2_b WM_BENCH_SLOTS=2
run_bench s3_b WM_BENCH_SLOTS=3
run_bench s4_b WM_BENCH_SLOTS=4
echo "== summary ($SECONDS s) =="
python - <<'PY'
import json, glob, os
out = os.environ["OUT"]
for f in sorted(glob.glob(os.path.join(out, "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        h = d["host"]
        print("%-12s %.4f Gbp/s ms/step %5.0f cpu_s %.1f util %.2f calls %s wall %s idle %.0f hits %d" % (os.path.basename(f)[6:-5], d["value"], d["ms_per_step"], h["process_cpu_s"], h["cpu_utilisation"], h["batched_calls"], h["batched_wall_s"], h["idle_wall_s"], d["config"]["hits"]))
    except Exception as e:
        print(f, "unreadable:", e)
PY
# host glue per named region (WM_PROF=1)
( WM_PROF=1 WM_BENCH_FILE=0 WM_BENCH_CPU_SAMPLE=0 timeout 150 python bench.py --steps 4 --warmup 2 --reads-per-step 16384 > $OUT/bench_prof.json 2> $OUT/bench_prof.log ); grep "\[prof\]" $OUT/bench_prof.log | sort -k3 -n -r | head -30
