#!/bin/bash
# round 6, GPU call aa: dense chain fill with a mask-free path for tiles wholly inside the scan and 32-bit tile bounds: parity, time per anchor, config 5 at size
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06aa
timeout 1200 python -m pytest tests/test_aux_gpu.py tests/test_window_gpu.py -x -q -m gpu 2>&1 | tail -2
python tools/chain_fill_check.py 16x5 8x5 4x10 16x3 2>&1 | cut -c1-200
python tools/chain_fill_probe.py 2>&1 | cut -c1-200 | head -12 | tee gpurun_out/r06aa/chain_fill_probe.txt
for v in a b; do
WM_TRACE=1 timeout 900 python tools/closure_run.py config5 --contigs 200 --ref-mb 3000 --skip-ref --out gpurun_out/r06aa/c5.json > gpurun_out/r06aa/c5_$v.log 2>&1
grep "window n=\|mapped" gpurun_out/r06aa/c5_$v.log | tail -3
done
