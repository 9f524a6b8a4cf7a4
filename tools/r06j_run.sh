#!/bin/bash
# round 6, GPU call j: the dense chain fill with the whole predecessor window per step (chain_block_wide) on a contig across a satellite array, then BASELINE
# config 5 at r05's size (24 x 5 Mb vs 600 Mb) and at its STATED size (200 x 5 Mb vs 3 Gb, asm20)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06j; mkdir -p $O
WM_CHAIN_WIDE=0 timeout 900 python tools/satellite_probe.py > $O/sat_wide0.txt 2>&1; echo "sat wide0 rc=$? $(tail -1 $O/sat_wide0.txt)"
WM_CHAIN_WIDE=1 timeout 900 python tools/satellite_probe.py > $O/sat_wide1.txt 2>&1; echo "sat wide1 rc=$? $(tail -1 $O/sat_wide1.txt)"
timeout 900 python tools/closure_run.py config5 --out $O/closure.jsonl > $O/c5_small.json 2> $O/c5_small.log; echo "config5 24x5Mb/600Mb rc=$? $(python -c "import json; d=json.load(open('$O/c5_small.json')); print(d['map_seconds'], d['reference_binary_seconds'], d['parity']['mismatches'])")"
timeout 2400 python tools/closure_run.py config5 --contigs 200 --ref-mb 3000 --arena-gb 60 --out $O/closure.jsonl > $O/c5_full.json 2> $O/c5_full.log; echo "config5 200x5Mb/3Gb rc=$? $(python -c "import json; d=json.load(open('$O/c5_full.json')); print(d['map_seconds'], d['reference_binary_seconds'], d['parity']['mismatches'])")"; tail -5 $O/c5_full.log
