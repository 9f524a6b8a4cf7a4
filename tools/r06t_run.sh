#!/bin/bash
# round 6, GPU call t: the race in chain_block_wide — which ingredient (merged barrier, conditional score store)
cd "$(dirname "$0")/.." || exit 1
for lib in libwmgpu libwmgpu_nomerge libwmgpu_alwayssc; do
  echo "#### $lib"
  WM_LIBWMGPU=$PWD/winnowmap_amd/$lib.so python tools/chain_fill_check.py 16x5 8x5 16x5 8x5 16x5 8x5 4x10 2>&1 | cut -c1-260
done
