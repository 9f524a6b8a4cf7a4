#!/bin/bash
# round 6, GPU call gg: glibc fastbins off (free() of a >= 64-KB table runs malloc_consolidate over the fastbins: ~10 % of the host's CPU samples) — host CPU seconds and throughput
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06gg; mkdir -p $O
export WM_BENCH_CACHE=/tmp/wmcache WM_BENCH_FILE=0
run() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 6 --warmup 2 --reads-per-step 32768 --cpu-sample 0 > $O/$name.json 2> $O/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(round(d['value'],4), d['host']['process_cpu_s'], d['host']['glue_cpu_s'])" 2>&1 | tail -1)"; }
run base1
run mxfast0_1 GLIBC_TUNABLES=glibc.malloc.mxfast=0
run base2
run mxfast0_2 GLIBC_TUNABLES=glibc.malloc.mxfast=0
run tcache0 GLIBC_TUNABLES=glibc.malloc.tcache_count=0
run arena4 GLIBC_TUNABLES=glibc.malloc.arena_max=4
