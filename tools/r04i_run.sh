#!/bin/bash
# GPU call r04i: validation of the round's host-side and index work on the GPU: aux (device index table), e2e (multi-mapper file loop, RCCL 1-rank), binding
# (MAPQ below the gate at 600 reads, split-prefix through the CLI), format.
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/r04i
mkdir -p $OUT
timeout 150 python -m pytest tests/test_aux_gpu.py -m gpu -x -q -k "index_build_on_device or refuses or kmer" > $OUT/gputest_aux.txt 2>&1; echo "aux rc=$? $SECONDS s"; tail -4 $OUT/gputest_aux.txt
timeout 200 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q > $OUT/gputest_e2e.txt 2>&1; echo "e2e rc=$? $SECONDS s"; tail -4 $OUT/gputest_e2e.txt
timeout 400 python -m pytest tests/test_binding_gpu.py -m gpu -x -q -k "bound_to_the_library or below_the_mcas or split_prefix or splice_mode_of_the_mapper or junction" > $OUT/gputest_binding.txt 2>&1; echo "binding rc=$? $SECONDS s"; tail -6 $OUT/gputest_binding.txt
echo "== done ($SECONDS s) =="
