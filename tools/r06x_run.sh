#!/bin/bash
# round 6, GPU call x: narrow first step of the dense chain fill: probe, parity, config 5 at size with it / with the round-5 fill
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06x
python tools/chain_fill_probe.py 2>&1 | cut -c1-200 | tee gpurun_out/r06x/chain_fill_probe.txt
python tools/chain_fill_check.py 16x5 8x5 4x10 2>&1 | cut -c1-200
for v in first1 wide0; do
  if [ $v = wide0 ]; then export WM_CHAIN_WIDE=0; fi
  WM_TRACE=1 timeout 900 python tools/closure_run.py config5 --contigs 200 --ref-mb 3000 --skip-ref --out gpurun_out/r06x/c5_$v.json > gpurun_out/r06x/c5_$v.log 2>&1
  echo "## $v"; grep "window n=\|mapped" gpurun_out/r06x/c5_$v.log | tail -4
done
