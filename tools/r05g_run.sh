#!/bin/bash
# round 5, GPU call g: which kernel owns the 14-s window call of a 5-Mb contig (rocprofv3 --kernel-trace --stats on 3 contigs); per-operation trace of a config-2 run
cd "$(dirname "$0")/.." || exit 1
O=$PWD/gpurun_out/r05g; mkdir -p $O
export TMPDIR=/tmp
ROOT=$PWD
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o c5 -- python $ROOT/tools/closure_run.py config5 --contigs 3 --ref-mb 200 > $O/c5_prof.json 2> $O/c5_prof.log ); echo "c5 prof rc=$? t=$SECONDS"
find $O/prof_c5 -name "*kernel_stats.csv" | head -2; for f in $(find $O/prof_c5 -name "*kernel_stats.csv"); do head -25 $f | cut -c1-220; cp $f $O/c5_kernel_stats.csv; done
find $O/prof_c5 -name "*.db" -delete; find $O/prof_c5 -name "*.csv" -size +1M -delete
export WM_BENCH_CACHE=/tmp/wmcache WM_BENCH_FILE=0
WM_TRACE=1 timeout 240 python bench.py --steps 4 --warmup 2 --reads-per-step 16384 --cpu-sample 0 > $O/trace.json 2> $O/trace.log; echo "trace rc=$? t=$SECONDS"
grep "\[ops\|\[host\|\[site\|\[map_reads" $O/trace.log | tail -40 > $O/trace_summary.txt; grep "\[batch\]" $O/trace.log | awk '{k=$4; n[k]++; ms[k]+=$(NF-1); nr[k]+=substr($5,3)} END {for (k in n) print k, "calls", n[k], "ms", ms[k], "avg", ms[k]/n[k], "reqs", nr[k]}' >> $O/trace_summary.txt; cat $O/trace_summary.txt | cut -c1-250; rm -f $O/trace.log
du -sh $O
