#!/bin/bash
# round 6, GPU call cc: the query code of a row from a rotating register (WM_KSW_QROT=1, this build) against the code cache + window test + v_readlane (variant library):
# the ksw GPU suite, then A/B at 32 768 reads per step. Result: no gain (profiles/r06_sched.txt) — the change (commit history: not merged) is NOT in the tree; the script is kept as the record of the call
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06cc; mkdir -p $O
timeout 1500 python -m pytest tests/test_ksw_gpu.py -x -q -m gpu 2>&1 | tail -3
export WM_BENCH_CACHE=/tmp/wmcache WM_BENCH_FILE=0
run() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 6 --warmup 2 --reads-per-step 32768 --cpu-sample 0 > $O/$name.json 2> $O/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(round(d['value'],4), d['host']['process_cpu_s'])" 2>&1 | tail -1)"; }
run qrot1
run old1 WM_LIBWMGPU=$PWD/winnowmap_amd/libwmgpu_qrot0.so
run qrot2
run old2 WM_LIBWMGPU=$PWD/winnowmap_amd/libwmgpu_qrot0.so
run qrot3
run old3 WM_LIBWMGPU=$PWD/winnowmap_amd/libwmgpu_qrot0.so
python - <<'P'
import json
for n in ('qrot1','old1'):
    d=json.load(open('gpurun_out/r06cc/%s.json'%n))
    for k,v in sorted(d['roofline']['classes'].items(), key=lambda kv:-kv[1]['ms'])[:10]: print(n, "%-42s ms %7.0f cells %.3e gcups_res %6.1f union %6.1f" % (k, v['ms'], v['cells'], v['gcups_residency'], v['gcups_union']))
P
