#!/bin/bash
# round 5, GPU call l: the stripe kernel with split-phase LDS loads (this commit) against the library before it, wave priority 3 for the stripe classes, and
# fuller / concurrent launches of the huge and heavy queues — BASELINE config 2 at 32 768 reads per step, one variant per run, compare INSIDE the call;
# the isolated wide-hull probe on both libraries; the ksw GPU tests and the split-index test on the new library
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05l; mkdir -p $O
export PYTHONFAULTHANDLER=1 TMPDIR=/tmp
export WM_BENCH_CACHE=/tmp/wmcache WM_BENCH_FILE=0
BASE=$PWD/winnowmap_amd/libwmgpu_base.so; PRIO3=$PWD/winnowmap_amd/libwmgpu_prio3.so
run() { name=$1; shift; env "$@" timeout 200 python bench.py --steps 4 --warmup 2 --reads-per-step 32768 --cpu-sample 0 > $O/$name.json 2> $O/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(round(d['value'],4), d['host']['process_cpu_s'], d['host']['cpu_quota_throttled'])" 2>&1 | tail -1) t=$SECONDS"; }
run base_a WM_LIBWMGPU=$BASE
run new_a
run prio3_a WM_LIBWMGPU=$PRIO3
run kx2048 WM_KSWX_MIN_BATCH=2048 WM_KSWX_MAX_WAIT_MS=600
run kx1024 WM_KSWX_MIN_BATCH=1024 WM_KSWX_MAX_WAIT_MS=400
run kx2 WM_KSWX_MAX=2 WM_KSWX_MIN_MORE=512
run kh24k WM_KSWH_MIN_BATCH=24576 WM_KSWH_MAX_WAIT_MS=120
run huge256k WM_KSW_HUGE_UNITS=262144
run base_b WM_LIBWMGPU=$BASE
run new_b
run prio3_b WM_LIBWMGPU=$PRIO3
WM_LIBWMGPU=$BASE timeout 60 python tools/ksw_probe.py 2000 > $O/probe_base.txt 2>&1; echo "probe base rc=$? t=$SECONDS"
timeout 60 python tools/ksw_probe.py 2000 > $O/probe_new.txt 2>&1; echo "probe new rc=$? t=$SECONDS"
grep -h "p16_1500\|blk_3000\|ext_5000\|blk2_6000" $O/probe_base.txt | cut -c1-150; echo --; grep -h "p16_1500\|blk_3000\|ext_5000\|blk2_6000" $O/probe_new.txt | cut -c1-150
timeout 400 python -m pytest tests/test_ksw_gpu.py -x -q -m gpu > $O/ksw_tests.log 2>&1; echo "ksw tests rc=$? t=$SECONDS"; tail -3 $O/ksw_tests.log
timeout 300 python -m pytest tests/test_binding_gpu.py -x -q -m gpu -k "parts" > $O/split_test.log 2>&1; echo "split test rc=$? t=$SECONDS"; tail -5 $O/split_test.log
du -sh $O
