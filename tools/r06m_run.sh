#!/bin/bash
# round 6, GPU call m: the headline workload under rocprofv3 (kernel table + timeline) with the per-call account of the host's hub (WM_TRACE), PMC passes included
cd "$(dirname "$0")/.." || exit 1
export WM_BENCH_CACHE=/tmp/wmcache WM_BENCH_FILE=0
PMC=1 SKIP_PLAIN=1 PROF_ARGS="--cpu-sample 0" bash tools/prof_bench.sh r06_mid --steps 4 --warmup 2 > gpurun_out/r06m_prof.log 2>&1; tail -45 gpurun_out/r06m_prof.log | cut -c1-200
WM_TRACE=1 timeout 600 python bench.py --steps 4 --warmup 2 --cpu-sample 0 > gpurun_out/r06m_trace.json 2> gpurun_out/r06m_trace.log; grep "\[ops\|\[map_reads\]\|\[host\]" gpurun_out/r06m_trace.log | tail -12 | cut -c1-300
grep "\[batch\]" gpurun_out/r06m_trace.log | awk '{k=$4; n[k]++; ms[k]+=$6; req[k]+=substr($5,3)} END {for (k in n) printf "%-12s calls %5d  avg %.1f ms  avg batch %d\n", k, n[k], ms[k]/n[k], req[k]/n[k]}'
gzip -f gpurun_out/r06m_trace.log
