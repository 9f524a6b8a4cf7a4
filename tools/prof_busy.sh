#!/bin/bash
# usage: tools/prof_busy.sh <tag> <n_reads> <ref_mb> <W|-> <threads>: kernel trace of the mapper probe + GPU busy-time analysis
TAG=$1; N=$2; MB=$3; WOPT=$4; TH=$5
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
ROOT=$PWD
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/stats -o map -- python $ROOT/tools/map_probe.py $N $MB $WOPT $TH > $OUT/probe.log 2>&1 )
grep -v '^W2026' $OUT/probe.log | grep "batch of" 
python3 - <<PY
import sqlite3,glob,collections
for f in glob.glob("$OUT/stats/*.db"):
    db=sqlite3.connect(f)
    rows=list(db.execute("select name,start,end from kernels order by start"))
    t0=rows[0][1]; t1=max(r[2] for r in rows)
    # union of busy intervals
    busy=0; cur_s=None; cur_e=None
    for n,s,e in rows:
        if cur_e is None or s>cur_e:
            if cur_e is not None: busy+=cur_e-cur_s
            cur_s,cur_e=s,e
        else: cur_e=max(cur_e,e)
    busy+=cur_e-cur_s
    print("kernels: %d, span %.2f s, GPU busy (>=1 kernel running) %.2f s"%(len(rows),(t1-t0)/1e9,busy/1e9))
    agg=collections.defaultdict(lambda:[0,0.0])
    for n,s,e in rows: agg[n.split('(')[0][:40]][0]+=1; agg[n.split('(')[0][:40]][1]+=(e-s)/1e9
    for k,v in sorted(agg.items(), key=lambda x:-x[1][1])[:10]: print("  %-42s calls=%6d sum=%.2f s"%(k,v[0],v[1]))
PY
