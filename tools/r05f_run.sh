#!/bin/bash
# round 5, GPU call f: large window jobs with tied keys sorted on the host (window tests, config 5 again with per-call trace, config 2 / 3 A/B)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05f; mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_window_gpu.py tests/test_aux_gpu.py -x -q -m gpu > $O/window_tests.log 2>&1; echo "window tests rc=$? t=$SECONDS"; tail -3 $O/window_tests.log
WM_TRACE=1 timeout 600 python tools/closure_run.py config5 --contigs 12 --out $O/closure.jsonl > $O/config5.json 2> $O/config5.log; echo "config5 rc=$? t=$SECONDS"; grep closure $O/config5.log | tail -3
grep "\[batch\]" $O/config5.log | sort -k6 -n -r -t' ' | awk '{print $NF, $0}' | sort -n -r | head -12 | cut -d' ' -f2- ; grep -v "\[batch\]" $O/config5.log > $O/config5_short.log; rm -f $O/config5.log
WM_WINDOW_TIES_HOST=0 timeout 600 python tools/closure_run.py config5 --contigs 12 > $O/config5_dev.json 2> $O/config5_dev.log; echo "config5(device ties) rc=$? t=$SECONDS"; grep closure $O/config5_dev.log | tail -2
export WM_BENCH_CACHE=/tmp/wmcache WM_BENCH_FILE=0
run() { name=$1; shift; env "$@" timeout 240 python bench.py --steps 8 --warmup 4 --reads-per-step 16384 --cpu-sample 0 > $O/$name.json 2> $O/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(round(d['value'],4), d['host']['process_cpu_s'], d['host']['cpu_quota_throttled'])" 2>&1 | tail -1) t=$SECONDS"; }
run c2_host_a
run c2_dev_a WM_WINDOW_TIES_HOST=0
run c2_host_b
run c2_dev_b WM_WINDOW_TIES_HOST=0
run3() { name=$1; shift; env "$@" timeout 300 python bench.py --config 3 --steps 4 --warmup 2 --reads-per-step 8192 --cpu-sample 0 > $O/$name.json 2> $O/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(round(d['value'],4), d['host']['process_cpu_s'], d['host']['cpu_quota_throttled'])" 2>&1 | tail -1) t=$SECONDS"; }
run3 c3_host_a
run3 c3_dev_a WM_WINDOW_TIES_HOST=0
run3 c3_host_b
du -sh $O
