#!/bin/bash
# round 6, GPU call l: the whole GPU suite on the current build, the headline bench at full size (with its rocprofv3 summary), BASELINE config 4 as a bench line,
# BASELINE config 5 at its stated size with the reference's mapping phase timed beside it
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
O=$ROOT/gpurun_out/r06l; mkdir -p $O
export TMPDIR=/tmp WM_BENCH_CACHE=/tmp/wmcache
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; echo "gpu tests rc=$? $(tail -1 $O/gputests.txt)"
timeout 900 python bench.py --steps 4 --warmup 2 > $O/bench.json 2> $O/bench.log; echo "bench rc=$? $(python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['parity']['mismatches'], d['parity']['mapq_compared'], d['cpu_baseline']['value'])")"
timeout 2400 python tools/closure_run.py config5 --contigs 200 --ref-mb 3000 --arena-gb 60 --out $O/closure.jsonl > $O/c5_full.json 2> $O/c5_full.log; echo "config5 200x5Mb/3Gb rc=$? $(python -c "import json; d=json.load(open('$O/c5_full.json')); print(d['map_seconds'], d['reference_binary_seconds'], d['reference_mapping_seconds'], d['parity']['mismatches'])")"
timeout 2400 python bench.py --config 4 --steps 4 --warmup 2 > $O/bench_config4.json 2> $O/bench_config4.log; echo "bench config 4 rc=$? $(python -c "import json; d=json.load(open('$O/bench_config4.json')); print(d['value'], d['ms_per_step'], d['parity']['mismatches'], d['cpu_baseline']['value'])")"
