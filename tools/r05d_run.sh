#!/bin/bash
# round 5, GPU call d: where the closure script died (fault handler on), smoke(), the close path of the GPU tests
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05d; mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 300 python tools/closure_run.py config4 --gb 0.24 --reads 64 > $O/c4small.json 2> $O/c4small.log; echo "c4small rc=$? t=$SECONDS"; grep -v amdgpu.ids $O/c4small.log | tail -40
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$? t=$SECONDS"; tail -5 $O/smoke.log
timeout 600 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu > $O/e2e.log 2>&1; echo "e2e rc=$? t=$SECONDS"; tail -15 $O/e2e.log
