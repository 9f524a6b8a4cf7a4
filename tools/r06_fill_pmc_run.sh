#!/bin/bash
# round 6: what bounds the dense chain fill (chain_kernel_wide<5>, 16 wavefronts, the whole 5 000-predecessor window of every anchor scored): instruction counts and
# issue / wait cycles of the one workgroup alone on the chip (tools/chain_fill_probe.py's 256 000-anchor satellite set)
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
O=$ROOT/gpurun_out/r06_fill_pmc; mkdir -p $O
export TMPDIR=/tmp WM_CHAIN_WIDE_GEOM=16x5 WM_CHAIN_WIDE_FIRST=5
W="python $ROOT/tools/chain_fill_probe.py one"
( cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc1 -o k -- $W > $O/p1.log 2>&1 ); echo "pmc1 rc=$?"
( cd /tmp && rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH -d $O/pmc2 -o k -- $W > $O/p2.log 2>&1 ); echo "pmc2 rc=$?"
python tools/pmc_kernel.py $O chain_kernel > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt; tail -3 $O/p1.log
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
