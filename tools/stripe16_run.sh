#!/bin/bash
# Sixteen-wavefront stripe geometries (<1,16>, <2,16>; opt-in): GPU parity of the stripe tests with that routing, then the isolated probe with and without
# them on the WM_STRIPE_TIMING variant (cycles per phase).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/stripe16
timeout 40 python -m pytest tests/test_ksw_gpu.py -m gpu -x -q -k "stripe and nwv16" > gpurun_out/stripe16/gputest.txt 2>&1; echo "test rc=$? $SECONDS s"; tail -2 gpurun_out/stripe16/gputest.txt
WM_KSW_STRIPE16=1 WM_LIBWMGPU=$PWD/winnowmap_amd/libwmgpu_timing.so timeout 25 python tools/ksw_probe.py 2000 > gpurun_out/stripe16/probe16.txt 2>&1; echo "probe16 rc=$? $SECONDS s"
grep -v "^ont\|^library" gpurun_out/stripe16/probe16.txt | cut -c1-420
