#!/bin/bash
# round 5, last GPU call: the whole GPU suite + smoke on the final build (packed resident reads, -H, one index part at a time, split-phase stripe control words);
# profiles (PMC passes, rocprofv3 --kernel-trace --stats, timeline; raw databases deleted before the merge); bench.py with its default legs (file_to_file,
# cpu_baseline, parity) at the full step size
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
ROOT=$PWD
O=$ROOT/gpurun_out/r05n; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$? t=$SECONDS"; tail -3 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$? t=$SECONDS"; tail -1 $O/smoke.log
SKIP_PLAIN=1 PMC=1 PMC_READS=1024 timeout 420 bash tools/prof_bench.sh r05_last_bench --steps 6 --warmup 2 --reads-per-step 16384 > $O/prof_bench.log 2>&1; echo "prof rc=$? t=$SECONDS"; tail -5 $O/prof_bench.log | cut -c1-200
timeout 400 python bench.py --steps 4 --warmup 2 > $O/bench_default.json 2> $O/bench_default.log; echo "bench rc=$? t=$SECONDS"; cut -c1-200 $O/bench_default.json; grep "^\[bench\]" $O/bench_default.log | tail -6
du -sh gpurun_out/r05n gpurun_out/prof_r05_last_bench
