#!/bin/bash
# round 6, GPU call o: one more wavefront per SIMD for the kernels just above 128 registers (WM_OCC_HINT variant) vs the default build, full-size steps; config 5 with
# the giant multi-tile chain jobs on the workgroup kernel
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06o; mkdir -p $O
export WM_BENCH_CACHE=/tmp/wmcache WM_BENCH_FILE=0
run() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 4 --warmup 2 --cpu-sample 0 > $O/$name.json 2> $O/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(round(d['value'],4), d['host']['process_cpu_s'], d['host']['cpu_quota_throttled'])" 2>&1 | tail -1)"; }
run base1
run occ1 WM_LIBWMGPU=$PWD/winnowmap_amd/libwmgpu_occ.so
run base2
run occ2 WM_LIBWMGPU=$PWD/winnowmap_amd/libwmgpu_occ.so
WM_TRACE=1 timeout 1500 python tools/closure_run.py config5 --contigs 48 --ref-mb 3000 --arena-gb 60 --skip-ref > $O/c5_48.json 2> $O/c5_48.log; echo "c5 48 rc=$? $(python -c "import json; d=json.load(open('$O/c5_48.json')); print(d['map_seconds'])")"; grep "\[batch\]" $O/c5_48.log | awk '$6 > 300 {print}' | head
