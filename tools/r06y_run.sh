#!/bin/bash
# round 6, GPU call y: host CPU sampling profile of the headline workload on this build
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06y; mkdir -p $O
g++ -O2 -fPIC -shared -o tools/sprof/libsprof.so tools/sprof/sprof.cpp -ldl
ls tools/sprof/*.so
SPROF_MARK=1 SPROF_OUT=$O/base.sprof LD_PRELOAD=$PWD/tools/sprof/libsprof.so timeout 400 python bench.py --steps 8 --warmup 4 --reads-per-step 16384 --cpu-sample 0 > $O/base.json 2> $O/base.log
python tools/sprof/resolve.py $(ls $O/base.sprof.* | head -1) 60 > $O/base_sprof.txt 2>&1
python -c "import json; d=json.load(open('$O/base.json')); print(round(d['value'],4), d['host'])"
head -75 $O/base_sprof.txt
