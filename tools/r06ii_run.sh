#!/bin/bash
# round 6, GPU call ii: unclipped exact 8-pair extensions in the clipped exact 8-pair class (one launch instead of two per batched call) vs a class of their own
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06ii; mkdir -p $O
timeout 900 python -m pytest tests/test_ksw_gpu.py -x -q -m gpu 2>&1 | tail -2
export WM_BENCH_CACHE=/tmp/wmcache WM_BENCH_FILE=0
run() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 6 --warmup 2 --reads-per-step 32768 --cpu-sample 0 > $O/$name.json 2> $O/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(round(d['value'],4), d['host']['process_cpu_s'])" 2>&1 | tail -1)"; }
run merge1
run own1 WM_KSW_MERGE_P8X=0
run merge2
run own2 WM_KSW_MERGE_P8X=0
run merge3
run own3 WM_KSW_MERGE_P8X=0
