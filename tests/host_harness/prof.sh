#!/bin/bash
# host-glue profile (tests only): builds prof_main, maps $OUT/{ref.fa,rep.txt,reads.fa} (record pass on the oracle, then replays = the host glue alone)
#   N=400 THREADS=8 REPLAYS=5 tests/host_harness/prof.sh            CPU seconds of the glue per read / per Gbase
#   SPROF=1 ...                                                     + sampling profile of the replays (tools/sprof), symbolised
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=${OUT:-/tmp/hp}
g++ -std=c++17 -O2 -g -ffp-contract=off -w -o $OUT/prof_main $HERE/prof_main.cpp $HERE/../../oracle/wm_oracle.c -lz -pthread
cd $OUT
if [ -n "$SPROF" ]; then
	rm -f $OUT/sprof.txt.*
	SPROF_MARK=1 SPROF_OUT=$OUT/sprof.txt LD_PRELOAD=$HERE/../../tools/sprof/libsprof.so ./prof_main ref.fa rep.txt reads.fa ${N:-300} ${PRESET:-map-ont}
	python $HERE/../../tools/sprof/resolve.py $(ls $OUT/sprof.txt.* | head -1) ${TOP:-60}
else
	./prof_main ref.fa rep.txt reads.fa ${N:-300} ${PRESET:-map-ont}
fi
