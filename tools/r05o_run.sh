#!/bin/bash
# round 5, GPU call o (diagnostic, no code under test changes): the batched calls of a bench run, one line each (WM_TRACE=1), to see which queue's serial chain
# of calls covers how much of a step; the full step size, 2 timed + 1 warm-up steps
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05o; mkdir -p $O
export PYTHONFAULTHANDLER=1 TMPDIR=/tmp WM_BENCH_FILE=0
WM_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > $O/trace.json 2> $O/trace.log; echo "rc=$? t=$SECONDS"
grep "\[batch\]" $O/trace.log | awk '{k=$4; n[k]++; jobs[k]+=substr($5,3); ms[k]+=$6} END {for (k in n) printf "%-11s calls %5d  requests %9d  summed wall %9.1f ms  avg %7.1f ms  avg batch %8.0f\n", k, n[k], jobs[k], ms[k], ms[k]/n[k], jobs[k]/n[k]}' | sort > $O/batches.txt
cat $O/batches.txt
python -c "import json; d=json.load(open('$O/trace.json')); print('value', round(d['value'],4), 'ms_per_step', round(d['ms_per_step'],1), 'steps', d['steps'])" | tee -a $O/batches.txt
grep -v "\[batch\]" $O/trace.log | tail -5
grep "\[batch\]" $O/trace.log | gzip > $O/batch_lines.txt.gz; rm -f $O/trace.log
du -sh $O
