#!/bin/bash
# round 6, GPU call v: where config 5 at its stated size (200 x 5 Mb) spends its 41 s: batch trace, then a kernel trace
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06v
WM_TRACE=1 timeout 900 python tools/closure_run.py config5 --contigs 200 --ref-mb 3000 --skip-ref --out gpurun_out/r06v/c5.json > gpurun_out/r06v/c5.log 2>&1
grep -c . gpurun_out/r06v/c5.log
cd /tmp && export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06v/prof -o c5 -- python $GRAFT_REPO_ROOT/tools/closure_run.py config5 --contigs 200 --ref-mb 3000 --skip-ref > $GRAFT_REPO_ROOT/gpurun_out/r06v/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r06v/prof -name "*kernel_trace*" -size +1M -exec gzip -9 {} \;
find gpurun_out/r06v/prof -name "*.gz" -size +30M -delete
ls -la gpurun_out/r06v/prof/* | head
