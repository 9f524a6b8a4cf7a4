#!/bin/bash
# round 6, GPU call u: the dense chain fill geometries against the oracle with the parameter block laid out as the header has it; the aux suite
cd "$(dirname "$0")/.." || exit 1
python tools/chain_fill_check.py 0 16x5 8x5 16x5 8x5 4x10 8x10 16x3 2x10 2>&1 | cut -c1-200
timeout 900 python -m pytest tests/test_aux_gpu.py -x -q -m gpu 2>&1 | tail -3
