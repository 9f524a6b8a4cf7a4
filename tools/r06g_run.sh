#!/bin/bash
# round 6, GPU call g: where the group switch of the chained-workgroup kernels spends its time (need_group vs writing the group out), back-off 8 vs 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06g; mkdir -p $O
for v in timing timing_b0; do
WM_LIBWMGPU=$PWD/winnowmap_amd/libwmgpu_$v.so timeout 600 python tools/ksw_chain_probe.py 0.5 > $O/chain_probe_$v.txt 2>&1; echo "$v probe rc=$?"; grep "chain bp2" -A1 $O/chain_probe_$v.txt | grep -A1 "blk_3000\|s2_3001\|ext_5000" | cut -c1-330
done
