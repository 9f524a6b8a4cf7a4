#!/bin/bash
# round 6, GPU call hh: how the runtime moves the per-call tables and results (13 000 __amd_rocclr_copyBuffer blit kernels per 6 steps, 8 % of the summed kernel time): SDMA engines vs shader copies
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06hh; mkdir -p $O
export WM_BENCH_CACHE=/tmp/wmcache WM_BENCH_FILE=0
run() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 6 --warmup 2 --reads-per-step 32768 --cpu-sample 0 > $O/$name.json 2> $O/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(round(d['value'],4), d['host']['process_cpu_s'])" 2>&1 | tail -1)"; }
run base1
run sdma0 HSA_ENABLE_SDMA=0
run blit0 GPU_FORCE_BLIT_COPY_SIZE=0
run blit1m GPU_FORCE_BLIT_COPY_SIZE=1024
run base2
