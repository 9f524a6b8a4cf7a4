#!/bin/bash
# usage: tools/prof_ksw.sh <tag>   (run on the GPU box via gpurun) — rocprofv3 kernel stats + PMC passes for the ksw probe
set -x
TAG=${1:-r01}
cd $GRAFT_REPO_ROOT 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/stats -o ksw -- python $OLDPWD/tools/ksw_probe.py 20000 > $OUT/probe_stats.log 2>&1 )
( cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc1 -o ksw -- python $OLDPWD/tools/ksw_probe.py 20000 > $OUT/probe_pmc1.log 2>&1 )
( cd /tmp && rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU -d $OUT/pmc2 -o ksw -- python $OLDPWD/tools/ksw_probe.py 20000 > $OUT/probe_pmc2.log 2>&1 )
find $OUT -name "*.csv" | head -20
