#!/bin/bash
# round 6, GPU call b: first run of the chained-workgroup ksw kernels — the ksw GPU tests (incl. the new chain tests), the isolated probe chain vs stripe,
# then bench A/B at 16 384 reads per step: WM_KSW_CHAIN=0 (round-5 routing) | 1 (default) | 3 (+ long exact extensions)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06b; mkdir -p $O
export WM_BENCH_CACHE=/tmp/wmcache WM_BENCH_FILE=0
timeout 900 python -m pytest tests/test_ksw_gpu.py -m gpu -x -q > $O/ksw_tests.txt 2>&1; echo "ksw tests rc=$? $(tail -1 $O/ksw_tests.txt)"
timeout 600 python tools/ksw_chain_probe.py > $O/chain_probe.txt 2>&1; echo "probe rc=$?"; cat $O/chain_probe.txt | tail -40
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 8 --warmup 4 --reads-per-step 16384 --cpu-sample 0 > $O/$name.json 2> $O/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(round(d['value'],4), d['host']['process_cpu_s'], d['host']['cpu_quota_throttled'])" 2>&1 | tail -1)"; }
run chain0_a WM_KSW_CHAIN=0
run chain1_a WM_KSW_CHAIN=1
run chain3_a WM_KSW_CHAIN=3
run chain0_b WM_KSW_CHAIN=0
run chain1_b WM_KSW_CHAIN=1
run chain1_bp4 WM_KSW_CHAIN=1 WM_KSW_CHAIN_BP=4
run chain3_b WM_KSW_CHAIN=3
