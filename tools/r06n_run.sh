#!/bin/bash
# round 6, GPU call n: with the chained-workgroup kernels a huge / heavy call no longer fills compute units whole — more device contexts and more huge calls in flight, again
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06n; mkdir -p $O
export WM_BENCH_CACHE=/tmp/wmcache WM_BENCH_FILE=0
run() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 4 --warmup 2 --cpu-sample 0 > $O/$name.json 2> $O/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(round(d['value'],4), d['host']['process_cpu_s'], d['host']['cpu_quota_throttled'])" 2>&1 | tail -1)"; }
run base1
run ctx8 WM_CONTEXTS=8
run ctx10 WM_CONTEXTS=10 GPU_MAX_HW_QUEUES=24
run kswx2 WM_KSWX_MAX=2
run ctx8_kswx2 WM_CONTEXTS=8 WM_KSWX_MAX=2
run ctx8_kswx2_h3 WM_CONTEXTS=8 WM_KSWX_MAX=2 WM_KSWH_MAX=3
run base2
run ctx12_all WM_CONTEXTS=12 GPU_MAX_HW_QUEUES=28 WM_KSWX_MAX=2 WM_KSWH_MAX=3 WM_KSW_MAX=4 WM_WINDOW_MAX=4
