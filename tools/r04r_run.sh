#!/bin/bash
# GPU call r04r: the shipped library once more after the routing edit: stripe tests (both routings) and the end-to-end parity tests.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r04r
timeout 30 python -m pytest tests/test_ksw_gpu.py -m gpu -x -q -k "stripe" > gpurun_out/r04r/gputest_stripe.txt 2>&1; echo "stripe rc=$? $SECONDS s"; tail -2 gpurun_out/r04r/gputest_stripe.txt
timeout 40 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -k "not two_mappers" > gpurun_out/r04r/gputest_e2e.txt 2>&1; echo "e2e rc=$? $SECONDS s"; tail -2 gpurun_out/r04r/gputest_e2e.txt
