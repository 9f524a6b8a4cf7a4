#!/bin/bash
# round 5, GPU call j: long sequences sketched chunk by chunk (one wavefront per 16 384 positions): device index == host index, 1-Gbase capacity test,
# window tests, the asm20 parity test (1-Mb contigs), configs 4 / 5 again (index time, mapping time), the satellite contig
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05j; mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_aux_gpu.py tests/test_window_gpu.py tests/test_capacity_gpu.py "tests/test_binding_gpu.py::test_parity_at_scale_config5_shape_asm20" -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$? t=$SECONDS"; tail -4 $O/tests.log
timeout 900 python tools/closure_run.py config4 --out $O/closure.jsonl > $O/config4.json 2> $O/config4.log; echo "config4 rc=$? t=$SECONDS"; grep closure $O/config4.log | tail -4
timeout 600 python tools/closure_run.py config5 --out $O/closure.jsonl > $O/config5.json 2> $O/config5.log; echo "config5 rc=$? t=$SECONDS"; grep closure $O/config5.log | tail -3
timeout 300 python tools/satellite_probe.py > $O/sat.txt 2> $O/sat.log; echo "sat rc=$? t=$SECONDS"; cat $O/sat.txt
WM_SKETCH_LONG=0 timeout 300 python tools/satellite_probe.py > $O/sat_single.txt 2> $O/sat_single.log; echo "sat(single-wavefront sketch) rc=$? t=$SECONDS"; cat $O/sat_single.txt
du -sh $O
