#!/bin/bash
# usage: tools/prof_bench.sh <tag> [bench args...]
# 0) (PMC=1) separate rocprofv3 --pmc passes for the HBM traffic of the ksw kernels -> profiles/pmc_bytes_per_cell.json (read by bench.py),
# 1) plain bench.py run (the JSON line), 2) the same command under rocprofv3 --kernel-trace --stats (PROF_ARGS appended, default --cpu-sample 0),
# 3) GPU timeline. Everything lands in gpurun_out/prof_<tag>/; tools/prof_summary.py turns the .db files into profiles/<tag>.txt.
TAG=${1:-r01_bench}; shift
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
if [ "${PMC:-0}" = "1" ]; then
  # HBM traffic (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE, in their own passes, no trace domains)
  # (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; a reduced batch keeps the serialised counter runs short)
  ( cd /tmp && WM_BENCH_CPU_SAMPLE=0 WM_BENCH_FILE=0 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc1 -o bench -- python $ROOT/bench.py --steps 1 --warmup 0 --reads-per-step ${PMC_READS:-1024} > $OUT/pmc1.log 2>&1 )
  ( cd /tmp && WM_BENCH_CPU_SAMPLE=0 WM_BENCH_FILE=0 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc2 -o bench -- python $ROOT/bench.py --steps 1 --warmup 0 --reads-per-step ${PMC_READS:-1024} > $OUT/pmc2.log 2>&1 )
  # instruction issue: what bounds the DP kernels: wave-instructions by type, wave cycles (quad-cycles) and where they went
  ( cd /tmp && WM_BENCH_CPU_SAMPLE=0 WM_BENCH_FILE=0 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc3 -o bench -- python $ROOT/bench.py --steps 1 --warmup 0 --reads-per-step ${PMC_READS:-1024} > $OUT/pmc3.log 2>&1 )
  ( cd $ROOT && python tools/pmc_ratio.py gpurun_out/prof_$TAG > $OUT/pmc_ratio.txt 2>&1; cat $OUT/pmc_ratio.txt; cp profiles/pmc_bytes_per_cell.json profiles/insts_per_cell.json $OUT/ )
fi
if [ "${SKIP_PLAIN:-0}" != "1" ]; then
python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.log
grep -v "^W2026" $OUT/bench.log | tail -6; cat $OUT/bench.json
fi
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python $ROOT/bench.py "$@" ${PROF_ARGS:---cpu-sample 0} > $OUT/bench_prof.json 2> $OUT/bench_prof.log )
cat $OUT/bench_prof.json
for f in $OUT/stats/*.db; do python $ROOT/tools/gpu_timeline.py $f > $OUT/timeline.txt 2>&1; done; cat $OUT/timeline.txt
cd $ROOT && python tools/prof_summary.py $TAG gpurun_out/prof_$TAG > /dev/null && cp profiles/$TAG.txt $OUT/ && head -30 profiles/$TAG.txt
# gpurun merges at most 64 MiB back: the raw databases stay on the box, the summaries travel
if [ "${KEEP_RAW:-0}" != "1" ]; then find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete; rm -rf $OUT/pmc1 $OUT/pmc2 $OUT/pmc3; fi
