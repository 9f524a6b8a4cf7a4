#!/bin/bash
# GPU call r04l: where the host's CPU seconds go: bench.py under the sampling profiler (tools/sprof), + the z-drop scan unit test.
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/r04l
mkdir -p $OUT
timeout 100 python -m pytest tests/test_ksw_gpu.py -m gpu -x -q -k "position_jobs" > $OUT/gputest_ksw.txt 2>&1; echo "ksw rc=$? $SECONDS s"; tail -2 $OUT/gputest_ksw.txt
g++ -O2 -fPIC -shared -o /tmp/libsprof.so tools/sprof/sprof.cpp -ldl
( SPROF_OUT=$OUT/sprof.txt LD_PRELOAD=/tmp/libsprof.so WM_BENCH_FILE=0 WM_BENCH_CPU_SAMPLE=0 timeout 200 python bench.py --steps 8 --warmup 2 --reads-per-step 16384 > $OUT/bench_sprof.json 2> $OUT/bench_sprof.log ); echo "bench rc=$? $SECONDS s"; cut -c1-120 $OUT/bench_sprof.json
ls -la $OUT/
for f in $OUT/sprof.txt.*; do python tools/sprof/resolve.py $f 60 > $f.resolved 2>&1; head -50 $f.resolved; done
