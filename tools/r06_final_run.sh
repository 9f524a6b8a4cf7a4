#!/bin/bash
# round 6, final measurements on the final build: GPU suite, PMC passes + headline bench + the same under rocprofv3 --kernel-trace --stats + timeline
# (tools/prof_bench.sh), BASELINE config 5 at its stated size with the reference beside it, BASELINE configs 4 and 3 as bench lines
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
O=$ROOT/gpurun_out/r06_final; mkdir -p $O
export TMPDIR=/tmp WM_BENCH_CACHE=/tmp/wmcache
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; echo "gpu tests rc=$? $(tail -1 $O/gputests.txt)"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" >> $O/gputests.txt 2>&1; tail -1 $O/gputests.txt
PMC=1 PROF_ARGS="--cpu-sample 0 --reads-per-step 16384 --steps 4 --warmup 2" timeout 2400 bash tools/prof_bench.sh r06_final_bench --steps 4 --warmup 2 > $O/prof_bench.log 2>&1; echo "prof_bench rc=$?"
python -c "import json; d=json.load(open('$ROOT/gpurun_out/prof_r06_final_bench/bench.json')); print('bench', d['value'], d['ms_per_step'], d['parity']['mismatches'], d['parity']['mapq_compared'], d['cpu_baseline']['value'])"
timeout 2400 python tools/closure_run.py config5 --contigs 200 --ref-mb 3000 --arena-gb 60 --out $O/closure.jsonl > $O/c5_full.json 2> $O/c5_full.log; echo "config5 200x5Mb/3Gb rc=$? $(python -c "import json; d=json.load(open('$O/c5_full.json')); print(d['map_seconds'], d['reference_binary_seconds'], d['reference_mapping_seconds'], d['parity']['mismatches'])")"
timeout 2400 python bench.py --config 4 --steps 4 --warmup 2 > $O/bench_config4.json 2> $O/bench_config4.log; echo "bench config 4 rc=$? $(python -c "import json; d=json.load(open('$O/bench_config4.json')); print(d['value'], d['ms_per_step'], d['parity']['mismatches'], d['cpu_baseline']['value'])")"
timeout 2400 python bench.py --config 3 --steps 4 --warmup 3 > $O/bench_config3.json 2> $O/bench_config3.log; echo "bench config 3 rc=$? $(python -c "import json; d=json.load(open('$O/bench_config3.json')); print(d['value'], d['ms_per_step'], d['parity']['mismatches'], d['cpu_baseline']['value'])")"
