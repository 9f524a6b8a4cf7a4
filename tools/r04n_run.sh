#!/bin/bash
# GPU call r04n (final build of the round): A) profiles: PMC passes (HBM bytes and instructions per cell per ksw class), plain bench, rocprofv3 --kernel-trace
# --stats of the same command, timeline;  B) the whole GPU suite + smoke.
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/r04n
mkdir -p $OUT
PMC=1 PMC_READS=1024 timeout 420 bash tools/prof_bench.sh r04n_bench --steps 6 --warmup 2 --reads-per-step 16384 > $OUT/prof_bench.log 2>&1; echo "prof rc=$? $SECONDS s"; tail -5 $OUT/prof_bench.log | cut -c1-300
timeout 420 python -m pytest tests -m gpu -x -q > $OUT/gputest.txt 2>&1; echo "gputest rc=$? $SECONDS s"; tail -4 $OUT/gputest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; echo "smoke rc=$? $SECONDS s"; tail -2 $OUT/smoke.txt
