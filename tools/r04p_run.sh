#!/bin/bash
# GPU call r04p: last check of the shipped library after the host-side change to the split-index flow (hits spilled to temporary files).
set -u
export TMPDIR=/tmp
export OUT=$PWD/gpurun_out/r04p
mkdir -p $OUT
timeout 90 python -m pytest tests/test_binding_gpu.py -m gpu -x -q -k "split_prefix" > $OUT/gputest_split.txt 2>&1; echo "split rc=$? $SECONDS s"; tail -2 $OUT/gputest_split.txt
timeout 60 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -k "not two_mappers" > $OUT/gputest_e2e.txt 2>&1; echo "e2e rc=$? $SECONDS s"; tail -2 $OUT/gputest_e2e.txt
