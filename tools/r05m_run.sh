#!/bin/bash
# round 5, GPU call m: the resident reads as 2 bits per base + ambiguity bitmap (csrc/reads2bit.h) — the GPU tests that touch resident reads, then the
# bench against the library of commit 238f6d7 (byte codes) in the same call
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05m; mkdir -p $O
export PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 500 python -m pytest tests/test_ksw_gpu.py tests/test_window_gpu.py tests/test_aux_gpu.py tests/test_e2e_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$? t=$SECONDS"; tail -5 $O/tests.log
export WM_BENCH_CACHE=/tmp/wmcache WM_BENCH_FILE=0
BASE=$PWD/winnowmap_amd/libwmgpu_base.so
run() { name=$1; shift; env "$@" timeout 200 python bench.py --steps 4 --warmup 2 --reads-per-step 32768 --cpu-sample 0 > $O/$name.json 2> $O/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(round(d['value'],4), d['host']['process_cpu_s'], d['parity']['mismatches'] if d.get('parity') else None)" 2>&1 | tail -1) t=$SECONDS"; }
run new_a
run base_a WM_LIBWMGPU=$BASE
run new_b
run base_b WM_LIBWMGPU=$BASE
du -sh $O
