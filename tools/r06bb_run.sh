#!/bin/bash
# round 6, GPU call bb: two alignments per wavefront (ksw_dual_kernel) — the ksw GPU suite, then A/B against one per wavefront (WM_KSW_DUAL=0) at 32 768 reads per step
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06bb; mkdir -p $O
timeout 1500 python -m pytest tests/test_ksw_gpu.py -x -q -m gpu 2>&1 | tail -4
export WM_BENCH_CACHE=/tmp/wmcache WM_BENCH_FILE=0
run() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 6 --warmup 2 --reads-per-step 32768 --cpu-sample 0 > $O/$name.json 2> $O/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(round(d['value'],4), d['host']['process_cpu_s'], d['parity']['mismatches'] if 'parity' in d else '-')" 2>&1 | tail -1)"; }
run dual1
run single1 WM_KSW_DUAL=0
run dual2
run single2 WM_KSW_DUAL=0
run dual3
python - <<'P'
import json
for n in ('dual1','single1'):
    d=json.load(open('gpurun_out/r06bb/%s.json'%n))
    for k,v in sorted(d['roofline']['classes'].items(), key=lambda kv:-kv[1]['ms'])[:12]: print(n, "%-42s ms %7.0f cells %.3e gcups_res %6.1f union %6.1f" % (k, v['ms'], v['cells'], v['gcups_residency'], v['gcups_union']))
P
