#!/bin/bash
# round 5, GPU call i: the whole GPU suite + smoke on the current build; reads in flight at the full step size (65 536 reads per step)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05i; mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$? t=$SECONDS"; tail -4 $O/gpu_tests.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$? t=$SECONDS"; tail -2 $O/smoke.log
export WM_BENCH_CACHE=/tmp/wmcache WM_BENCH_FILE=0
run() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 4 --warmup 2 --cpu-sample 0 > $O/$name.json 2> $O/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(round(d['value'],4), d['host']['process_cpu_s'], d['host']['cpu_quota_throttled'])" 2>&1 | tail -1) t=$SECONDS"; }
run full_a
run full_inflight32k WM_INFLIGHT=32768
run full_b
run full_s3 WM_BENCH_SLOTS=3 WM_BENCH_THREADS=12
du -sh $O
