#!/bin/bash
# round 6, last build: the whole GPU suite + smoke, the headline bench line
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
O=$ROOT/gpurun_out/r06_last; mkdir -p $O
export TMPDIR=/tmp WM_BENCH_CACHE=/tmp/wmcache
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; echo "gpu tests rc=$? $(tail -1 $O/gputests.txt)"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" >> $O/gputests.txt 2>&1; tail -1 $O/gputests.txt
timeout 900 python bench.py --steps 4 --warmup 2 > $O/bench.json 2> $O/bench.log; echo "bench rc=$? $(python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['parity']['mismatches'], d['parity']['mapq_compared'], d['cpu_baseline']['value'])")"
