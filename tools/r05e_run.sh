#!/bin/bash
# round 5, GPU call e: configs 4 / 5 at size (parity records), the new scheduling defaults against the old ones and against three mini-batches in flight,
# a host profile of config 3 (HiFi: host-bound), the ksw GPU tests on the build with the watchdog
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05e; mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 900 python tools/closure_run.py config4 --out $O/closure.jsonl > $O/config4.json 2> $O/config4.log; echo "config4 rc=$? t=$SECONDS"; grep closure $O/config4.log | tail -3
WM_TRACE=1 timeout 600 python tools/closure_run.py config5 --out $O/closure.jsonl > $O/config5.json 2> $O/config5.log; echo "config5 rc=$? t=$SECONDS"; grep closure $O/config5.log | tail -3
grep -c "\[batch\]" $O/config5.log; grep "\[batch\]" $O/config5.log | awk '{k=$4; n[k]++; ms[k]+=$(NF-1)} END {for (k in n) print k, n[k], ms[k]}' > $O/config5_batches.txt; cat $O/config5_batches.txt
grep -v "\[batch\]" $O/config5.log > $O/config5_short.log; grep "\[batch\]" $O/config5.log | sort -k6 -n -r -t' ' | head -40 > $O/config5_longest_batches.txt; rm -f $O/config5.log
export WM_BENCH_CACHE=/tmp/wmcache WM_BENCH_FILE=0
run() { name=$1; shift; env "$@" timeout 240 python bench.py --steps 8 --warmup 4 --reads-per-step 16384 --cpu-sample 0 > $O/$name.json 2> $O/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(round(d['value'],4), d['host']['process_cpu_s'], d['host']['cpu_quota_throttled'])" 2>&1 | tail -1) t=$SECONDS"; }
run new_a
run s3t12_a WM_BENCH_SLOTS=3 WM_BENCH_THREADS=12
run old_a WM_KSWX_MIN_BATCH=256 WM_KSWX_MAX_WAIT_MS=100 WM_KSW_STRIPE16=0
run new_b
run s3t12_b WM_BENCH_SLOTS=3 WM_BENCH_THREADS=12
run s3t10 WM_BENCH_SLOTS=3 WM_BENCH_THREADS=10
SPROF_MARK=1 SPROF_OUT=$O/c3.sprof LD_PRELOAD=$PWD/tools/sprof/libsprof.so timeout 300 python bench.py --config 3 --steps 4 --warmup 2 --reads-per-step 8192 --cpu-sample 0 > $O/c3_prof.json 2> $O/c3_prof.log; python tools/sprof/resolve.py $(ls $O/c3.sprof.* | head -1) 50 > $O/c3_sprof.txt 2>&1; rm -f $O/c3.sprof.*; echo "c3prof t=$SECONDS"; head -30 $O/c3_sprof.txt
timeout 600 python -m pytest tests/test_ksw_gpu.py -x -q -m gpu > $O/ksw_tests.log 2>&1; echo "ksw tests rc=$? t=$SECONDS"; tail -3 $O/ksw_tests.log
du -sh $O
