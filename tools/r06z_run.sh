#!/bin/bash
# round 6, GPU call z: 512-lane stripes for the chained-workgroup ksw classes (WM_KSW_CHAIN_BP=4) and the wider routings against the default, 32 768 reads per step
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06z; mkdir -p $O
export WM_BENCH_CACHE=/tmp/wmcache WM_BENCH_FILE=0
run() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 6 --warmup 2 --reads-per-step 32768 --cpu-sample 0 > $O/$name.json 2> $O/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(round(d['value'],4), d['host']['process_cpu_s'], d['parity']['mismatches'])" 2>&1 | tail -1)"; }
run base1
run bp4_1 WM_KSW_CHAIN_BP=4
run base2
run bp4_2 WM_KSW_CHAIN_BP=4
run rows1k WM_KSW_CHAIN_ROWS=1024
run chain0 WM_KSW_CHAIN=0
