#!/bin/bash
# round 6, GPU call h: instruction counts and wait cycles of the chained-workgroup kernel alone on the chip (one shape: 3001-wide band, 12 000 x 12 000, exact)
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
O=$ROOT/gpurun_out/r06h; mkdir -p $O
export TMPDIR=/tmp
W="python $ROOT/tools/ksw_one_shape.py 12000 32 40 3001"
( cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES -d $O/pmc1 -o k -- $W 3 2 > $O/p1.log 2>&1 ); echo "pmc1 rc=$?"
( cd /tmp && rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH -d $O/pmc2 -o k -- $W 3 2 > $O/p2.log 2>&1 ); echo "pmc2 rc=$?"
( cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc3 -o k -- $W 0 2 > $O/p3.log 2>&1 ); echo "pmc3 (stripe) rc=$?"
python tools/pmc_kernel.py $O ksw_ > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt; cat $O/p1.log | tail -2
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
