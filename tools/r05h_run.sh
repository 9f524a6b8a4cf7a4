#!/bin/bash
# round 5, GPU call h: a contig across a satellite array under rocprofv3 --kernel-trace --stats: which kernels own the window call
cd "$(dirname "$0")/.." || exit 1
O=$PWD/gpurun_out/r05h; mkdir -p $O
export TMPDIR=/tmp
ROOT=$PWD
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof -o sat -- python $ROOT/tools/satellite_probe.py > $O/sat.txt 2> $O/sat.log ); echo "rc=$? t=$SECONDS"; cat $O/sat.txt
for f in $(find $O/prof -name "*.db"); do python - "$f" > $O/sat_kernels.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
print("%-100s %8s %12s %12s %6s" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
for r in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 25"):
    print("%-100s %8d %12.1f %12.2f %6.2f" % (r[0][:100], r[1], r[2] / 1e6, r[3] / 1e6, r[4]))
PY
done; cat $O/sat_kernels.txt | cut -c1-150
find $O/prof -name "*.db" -delete; rm -rf $O/prof
