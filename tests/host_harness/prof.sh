#!/bin/bash
# host-glue profile (tests only): builds prof_main with -pg, maps /tmp/hp inputs (ref.fa rep.txt reads.fa), prints the gprof flat profile of the REPLAY pass
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=${OUT:-/tmp/hp}
g++ -std=c++17 -O2 -g ${PG:--pg} -ffp-contract=off -o $OUT/prof_main $HERE/prof_main.cpp $HERE/../../oracle/wm_oracle.c -lz -pthread
cd $OUT && ./prof_main ref.fa rep.txt reads.fa ${N:-300} ${PRESET:-map-ont}
