#!/bin/bash
# round 6, GPU call f: the prefetched group lands in accumulation registers the compiler does not see (no copies of a pending load, no wait outside the group switch)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06f; mkdir -p $O
export WM_BENCH_CACHE=/tmp/wmcache WM_BENCH_FILE=0
timeout 600 python -m pytest tests/test_ksw_gpu.py -m gpu -x -q -k chain > $O/ksw_tests.txt 2>&1; echo "ksw chain tests rc=$? $(tail -1 $O/ksw_tests.txt)"
WM_LIBWMGPU=$PWD/winnowmap_amd/libwmgpu_timing.so timeout 600 python tools/ksw_chain_probe.py 0.5 > $O/chain_probe_timing.txt 2>&1; echo "timing probe rc=$?"; grep "chain bp2" -A1 $O/chain_probe_timing.txt | head -60
timeout 600 python tools/ksw_chain_probe.py > $O/chain_probe.txt 2>&1; echo "probe rc=$?"; cat $O/chain_probe.txt | tail -34
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 8 --warmup 4 --reads-per-step 16384 --cpu-sample 0 > $O/$name.json 2> $O/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(round(d['value'],4), d['host']['process_cpu_s'], d['host']['cpu_quota_throttled'])" 2>&1 | tail -1)"; }
run chain0_a WM_KSW_CHAIN=0
run chain1_a WM_KSW_CHAIN=1
run chain3_a WM_KSW_CHAIN=3
run chain1_b WM_KSW_CHAIN=1
