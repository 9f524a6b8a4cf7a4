#!/bin/bash
# round 6, GPU call k: what the 42 s of BASELINE config 5 at its stated size are made of — rocprofv3 --stats of 48 contigs vs the 3-Gb reference + the host's account
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
O=$ROOT/gpurun_out/r06k; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && WM_TRACE=1 timeout 1500 rocprofv3 --kernel-trace --stats -d $O/prof -o c5 -- python $ROOT/tools/closure_run.py config5 --contigs 48 --ref-mb 3000 --arena-gb 60 --skip-ref > $O/c5_48.json 2> $O/c5_48.log ); echo "rc=$?"
python - <<'P'
import sqlite3, glob, os
f = glob.glob(os.path.join(os.environ.get("O", "gpurun_out/r06k"), "prof", "**", "*.db"), recursive=True)
db = sqlite3.connect(f[0])
rows = list(db.execute("select name, total_calls, total_duration, average from top_kernels"))
tot = sum(r[2] for r in rows) or 1
for nm, calls, total, avg in rows[:25]:
    print("%-100s %7d %10.1f ms %9.1f us %5.1f %%" % (nm[:100], calls, total / 1e6, avg / 1e3, 100.0 * total / tot))
P
grep -i "\[host\]\|\[ops\|\[map_reads\]\|closure" $O/c5_48.log | tail -12
python -c "
import json; d=json.load(open('$O/c5_48.json')); print(d['map_seconds'], d['host'])"
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
