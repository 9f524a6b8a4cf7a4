#!/bin/bash
# round 6, GPU call c: where a chained-workgroup wavefront's cycles go (WM_STRIPE_TIMING variant), and the probe again with the waits moved into the rare branches
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06c; mkdir -p $O
WM_LIBWMGPU=$PWD/winnowmap_amd/libwmgpu_timing.so timeout 600 python tools/ksw_chain_probe.py 0.5 > $O/chain_probe_timing.txt 2>&1; echo "timing probe rc=$?"; grep -v "^library" $O/chain_probe_timing.txt | head -80
timeout 600 python tools/ksw_chain_probe.py > $O/chain_probe.txt 2>&1; echo "probe rc=$?"; cat $O/chain_probe.txt | tail -40
timeout 600 python -m pytest tests/test_ksw_gpu.py -m gpu -x -q -k chain > $O/ksw_tests.txt 2>&1; echo "ksw chain tests rc=$? $(tail -1 $O/ksw_tests.txt)"
