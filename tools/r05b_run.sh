#!/bin/bash
# round 5, GPU call b: host-side work of the round (fast record formatting, pinned job tables, vector ksw_ll / update_extra scan, fused region selection)
# against the round-4 library in one call; huge / heavy queue batching; what the per-class events cost; host sampling profile with caller words
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05b; mkdir -p $O
export WM_BENCH_CACHE=/tmp/wmcache WM_BENCH_FILE=0
run() { name=$1; shift; env "$@" timeout 240 python bench.py --steps 8 --warmup 4 --reads-per-step 16384 --cpu-sample 0 > $O/$name.json 2> $O/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(round(d['value'],4), d['host']['process_cpu_s'], d['host']['cpu_quota_throttled'])" 2>&1 | tail -1)"; }
run new1
run r04lib WM_LIBWMGPU=$PWD/winnowmap_amd/libwmgpu_r04.so
run new2
run noev WM_KSW_CLASS_EVENTS=0
run kswx512 WM_KSWX_MIN_BATCH=512 WM_KSWX_MAX_WAIT_MS=250
run kswx_h WM_KSWX_MIN_BATCH=512 WM_KSWX_MAX_WAIT_MS=250 WM_KSWH_MAX_WAIT_MS=120 WM_KSWH_MIN_BATCH=24576
run kswx_s16 WM_KSWX_MIN_BATCH=512 WM_KSWX_MAX_WAIT_MS=250 WM_KSW_STRIPE16=1
run ctx8 WM_CONTEXTS=8
run slots3 WM_BENCH_SLOTS=3 WM_BENCH_THREADS=12
SPROF_MARK=1 SPROF_OUT=$O/sprof.txt LD_PRELOAD=$PWD/tools/sprof/libsprof.so timeout 240 python bench.py --steps 8 --warmup 4 --reads-per-step 16384 --cpu-sample 0 > $O/sprof_bench.json 2> $O/sprof_bench.log
python tools/sprof/resolve.py $(ls $O/sprof.txt.* | head -1) 60 > $O/sprof_report.txt 2>&1
ls $O | wc -l
