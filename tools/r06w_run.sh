#!/bin/bash
# round 6, GPU call w: 32-bit chain scoring: aux / window suites, the fill geometries (parity + time per anchor), config 5 at size
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06w
timeout 1200 python -m pytest tests/test_aux_gpu.py tests/test_window_gpu.py -x -q -m gpu 2>&1 | tail -3
python tools/chain_fill_check.py 0 16x5 8x5 2>&1 | cut -c1-200
python tools/chain_fill_probe.py 2>&1 | cut -c1-200 | tee gpurun_out/r06w/chain_fill_probe.txt
WM_TRACE=1 timeout 900 python tools/closure_run.py config5 --contigs 200 --ref-mb 3000 --skip-ref --out gpurun_out/r06w/c5.json > gpurun_out/r06w/c5.log 2>&1
grep "window n=\|mapped" gpurun_out/r06w/c5.log | tail -8
