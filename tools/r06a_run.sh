#!/bin/bash
# round 6, GPU call a: the round's first state on the GPU — MAPQ parity above the MCAS gate (tests + bench parity block), the GPU suite, one bench line
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06a; mkdir -p $O
export WM_BENCH_CACHE=/tmp/wmcache
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; echo "gpu tests rc=$? $(tail -1 $O/gputests.txt)"
timeout 900 python bench.py --steps 4 --warmup 2 > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"
python - <<'P'
import json
d=json.load(open('gpurun_out/r06a/bench.json'))
print(d['value'], d['ms_per_step'], d.get('parity'), d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)
P
