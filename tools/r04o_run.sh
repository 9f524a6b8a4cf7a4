#!/bin/bash
# GPU call r04o (final build): A) profiles (PMC passes, rocprofv3 --kernel-trace --stats, timeline; raw databases deleted before the merge — r04n lost its
# outputs to the 64-MiB limit), B) bench.py with its default legs (file_to_file, cpu_baseline, parity) at 65 536 reads per step, C) config 3 at its stated size.
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/r04o
mkdir -p $OUT
SKIP_PLAIN=1 PMC=1 PMC_READS=1024 timeout 300 bash tools/prof_bench.sh r04o_bench --steps 6 --warmup 2 --reads-per-step 16384 > $OUT/prof_bench.log 2>&1; echo "prof rc=$? $SECONDS s"; du -sh gpurun_out
timeout 270 python bench.py --steps 4 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.log; echo "bench rc=$? $SECONDS s"; cut -c1-160 $OUT/bench_default.json
WM_BENCH_FILE=0 WM_BENCH_CPU_SAMPLE=256 timeout 200 python bench.py --config 3 --steps 2 --warmup 1 > $OUT/bench_config3.json 2> $OUT/bench_config3.log; echo "config3 rc=$? $SECONDS s"; cut -c1-160 $OUT/bench_config3.json
du -sh gpurun_out
