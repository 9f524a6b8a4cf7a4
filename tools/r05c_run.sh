#!/bin/bash
# round 5, GPU call c: (1) what the WAIT_EVENTS ioctls are (sampling profiles without the per-class events / without interrupt signals), mini-batches in flight
# with fewer host threads each; (2) BASELINE configs 4, 5 and 3 at their reference sizes
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05c; mkdir -p $O
export WM_BENCH_CACHE=/tmp/wmcache WM_BENCH_FILE=0
run() { name=$1; shift; env "$@" timeout 240 python bench.py --steps 8 --warmup 4 --reads-per-step 16384 --cpu-sample 0 > $O/$name.json 2> $O/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(round(d['value'],4), d['host']['process_cpu_s'], d['host']['cpu_quota_throttled'])" 2>&1 | tail -1) t=$SECONDS"; }
prof() { name=$1; shift; env "$@" SPROF_MARK=1 SPROF_OUT=$O/$name.sprof LD_PRELOAD=$PWD/tools/sprof/libsprof.so timeout 240 python bench.py --steps 8 --warmup 4 --reads-per-step 16384 --cpu-sample 0 > $O/$name.json 2> $O/$name.log; python tools/sprof/resolve.py $(ls $O/$name.sprof.* | head -1) 25 > $O/${name}_sprof.txt 2>&1; echo "$name $(python -c "import json; d=json.load(open('$O/$name.json')); print(round(d['value'],4), d['host']['process_cpu_s'])" 2>&1 | tail -1) $(grep -m1 ioctl $O/${name}_sprof.txt) t=$SECONDS"; }
prof p_noev WM_KSW_CLASS_EVENTS=0
prof p_noirq HSA_ENABLE_INTERRUPT=0
run s3t12 WM_BENCH_SLOTS=3 WM_BENCH_THREADS=12
run s3t12_kx WM_BENCH_SLOTS=3 WM_BENCH_THREADS=12 WM_KSWX_MIN_BATCH=512 WM_KSWX_MAX_WAIT_MS=250 WM_KSW_STRIPE16=1
run s3t16 WM_BENCH_SLOTS=3 WM_BENCH_THREADS=16
run s4t10 WM_BENCH_SLOTS=4 WM_BENCH_THREADS=10
run s3t12_c8 WM_BENCH_SLOTS=3 WM_BENCH_THREADS=12 WM_CONTEXTS=8
run s2t12_noev WM_BENCH_THREADS=12 WM_KSW_CLASS_EVENTS=0
rm -rf /tmp/wmcache
# ---- closure: configs 4 and 5 (parity records), config 3 through bench.py with a full-step CPU baseline
timeout 900 python tools/closure_run.py config4 --out $O/closure.jsonl > $O/config4.json 2> $O/config4.log; echo "config4 rc=$? t=$SECONDS"; tail -3 $O/config4.log
timeout 600 python tools/closure_run.py config5 --out $O/closure.jsonl > $O/config5.json 2> $O/config5.log; echo "config5 rc=$? t=$SECONDS"; tail -3 $O/config5.log
WM_BENCH_CACHE= WM_BENCH_FILE=1 WM_BENCH_CPU_THREADS=16 WM_BENCH_DISTINCT_BATCHES=3 timeout 600 python bench.py --config 3 --steps 4 --warmup 3 > $O/bench_config3.json 2> $O/bench_config3.log; echo "config3 rc=$? t=$SECONDS"; tail -4 $O/bench_config3.log; cut -c1-300 $O/bench_config3.json
rm -f $O/*.sprof.*
du -sh $O
