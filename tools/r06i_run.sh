#!/bin/bash
# round 6, GPU call i: polls of the chained-workgroup kernels spaced out (0.5 .. 1.7 us between two polls of a waiting wavefront)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06i; mkdir -p $O
timeout 600 python tools/ksw_chain_probe.py > $O/chain_probe.txt 2>&1; echo "probe rc=$?"; grep "chain bp2\|stripe" $O/chain_probe.txt | tail -24
