#!/bin/bash
# usage: tools/prof_map.sh <tag> [n_reads] [ref_mb]  — rocprofv3 kernel stats for the whole-mapper probe
TAG=${1:-r01}; N=${2:-256}; MB=${3:-20}; WOPT=${4:-}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
ROOT=$PWD
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/stats -o map -- python $ROOT/tools/map_probe.py $N $MB $WOPT > $OUT/probe_stats.log 2>&1 )
grep -v '^W2026' $OUT/probe_stats.log | tail -4
python3 - <<PY
import sqlite3,glob
for f in glob.glob("$OUT/stats/*.db"):
    db=sqlite3.connect(f)
    for r in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 14"): print("%-70s calls=%d total_us=%.0f avg_us=%.0f pct=%.1f"%(r[0][:70],r[1],r[2],r[3],r[4]))
PY
