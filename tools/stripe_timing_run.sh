#!/bin/bash
# Where a stripe wavefront's cycles go: the isolated probe on the WM_STRIPE_TIMING variant library (built on the CPU side beforehand:
#   WM_KERNEL_DEFINES="WM_STRIPE_TIMING=1" python -c "from winnowmap_amd import build; build.build_gpu(out='winnowmap_amd/libwmgpu_timing.so')" ).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/stripe_timing
WM_LIBWMGPU=$PWD/winnowmap_amd/libwmgpu_timing.so timeout ${1:-50} python tools/ksw_probe.py 2000 > gpurun_out/stripe_timing/probe.txt 2>&1; echo "rc=$? $SECONDS s"
grep -v "^ont" gpurun_out/stripe_timing/probe.txt | cut -c1-700
