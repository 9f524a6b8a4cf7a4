#!/bin/bash
# round 5, GPU call a: scheduling A/B at 16 384 reads per step (one library; every variant is an environment switch) + a host sampling profile
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05a; mkdir -p $O
export WM_BENCH_CACHE=/tmp/wmcache WM_BENCH_FILE=0
run() { name=$1; shift; env "$@" timeout 240 python bench.py --steps 8 --warmup 4 --reads-per-step 16384 --cpu-sample 0 > $O/$name.json 2> $O/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(round(d['value'],4), d['host']['process_cpu_s'], d['host']['cpu_quota_throttled'])" 2>&1 | tail -1)"; }
run base1
run s16_1 WM_KSW_STRIPE16=1
run s16_3 WM_KSW_STRIPE16=3
run cu64 WM_CU_SPLIT=64
run cu32 WM_CU_SPLIT=32
run cu96 WM_CU_SPLIT=96
run kswx512 WM_KSWX_MIN_BATCH=512 WM_KSWX_MAX_WAIT_MS=250
run base2
run thr12 WM_BENCH_THREADS=12
run thr10x0 WM_BENCH_THREADS=10 WM_EXTRA_WORKERS=3
run inflight32k WM_INFLIGHT=32768
# host sampling profile of one run (timed region only)
SPROF_MARK=1 SPROF_OUT=$O/sprof.txt LD_PRELOAD=$PWD/tools/sprof/libsprof.so timeout 240 python bench.py --steps 8 --warmup 4 --reads-per-step 16384 --cpu-sample 0 > $O/sprof_bench.json 2> $O/sprof_bench.log
python tools/sprof/resolve.py $(ls $O/sprof.txt.* | head -1) 70 > $O/sprof_report.txt 2>&1
nm -D --defined-only /lib/x86_64-linux-gnu/libc.so.6 > /dev/null 2>&1
ls $O
