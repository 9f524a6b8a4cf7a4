#!/bin/bash
# round 6, GPU call p: the library as five translation units — the whole GPU suite; what the giant window call of config 5 is made of now (48 contigs under rocprofv3)
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
O=$ROOT/gpurun_out/r06p; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; echo "gpu tests rc=$? $(tail -1 $O/gputests.txt)"
( cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats -d $O/prof -o c5 -- python $ROOT/tools/closure_run.py config5 --contigs 48 --ref-mb 3000 --arena-gb 60 --skip-ref > $O/c5_48.json 2> $O/c5_48.log ); echo "rc=$?"
python - <<'P'
import sqlite3, glob, os
f = glob.glob(os.path.join("gpurun_out/r06p", "prof", "**", "*.db"), recursive=True)
db = sqlite3.connect(f[0])
rows = list(db.execute("select name, total_calls, total_duration, average from top_kernels"))
tot = sum(r[2] for r in rows) or 1
for nm, calls, total, avg in rows[:14]:
    print("%-100s %7d %10.1f ms %9.2f ms %5.1f %%" % (nm[:100], calls, total / 1e6, avg / 1e6, 100.0 * total / tot))
P
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
