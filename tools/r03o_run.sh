#!/bin/bash
# round 3, GPU call O: lean <CLIP, HASN> variants of ksw_pmulti_kernel<4,4> for the 16-pair classes (A/B by WM_KSW_PMULTI_LEAN), then the
# profiles of the final build: bench + rocprofv3 --kernel-trace --stats + the PMC traffic passes (tools/prof_bench.sh)
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/${TAG:-r03o}
mkdir -p $OUT
for v in 1 0; do
  ( WM_KSW_PMULTI_LEAN=$v timeout 240 python tools/ksw_probe.py 20000 > $OUT/ksw_probe_lean$v.txt 2>&1 ); echo "probe lean=$v rc=$?"; grep "p16\|blk" $OUT/ksw_probe_lean$v.txt
done
echo "== ksw + e2e GPU tests with the default build =="
timeout 900 python -m pytest tests/test_ksw_gpu.py tests/test_e2e_gpu.py -m gpu -q -x > $OUT/gputest.txt 2>&1; echo "rc=$? $SECONDS s"; tail -3 $OUT/gputest.txt
run_bench() { local tag=$1; shift; local t0=$SECONDS
  ( env ${WM_ENV:-WM_X=1} timeout 900 python bench.py "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $((SECONDS-t0))s $(tail -1 $OUT/bench_$tag.log | cut -c1-200)"; }
export WM_BENCH_DISTINCT_BATCHES=2 WM_BENCH_CPU_SAMPLE=0
WM_ENV="WM_KSW_PMULTI_LEAN=0" run_bench lean0 --steps 4 --warmup 2
WM_ENV="WM_KSW_PMULTI_LEAN=1" run_bench lean1 --steps 4 --warmup 2
unset WM_BENCH_DISTINCT_BATCHES WM_BENCH_CPU_SAMPLE
echo "== profiles of the final build =="
PMC=1 PMC_READS=4096 timeout 1200 bash tools/prof_bench.sh r03o_bench > $OUT/prof_bench.txt 2>&1; echo "prof rc=$? $SECONDS s"; tail -25 $OUT/prof_bench.txt
python - <<'PY'
import os, json, glob
out = os.environ["OUT"]
for f in sorted(glob.glob(os.path.join(out, "bench_*.json"))) + [os.path.join(os.path.dirname(out), "prof_r03o_bench", "bench.json")]:
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        h = d["host"]
        print("%-22s %.4f Gbp/s ms/step %5.0f steps %d cpu/step %.1f | parity %s | cpu %s" % (os.path.basename(f), d["value"], d["ms_per_step"], d["steps"], h["process_cpu_s"] / d["steps"],
              (d.get("parity") or {}).get("mismatches"), (d.get("cpu_baseline") or {}).get("value")))
        for k, v in sorted(d["roofline"]["classes"].items(), key=lambda kv: -kv[1]["ms"])[:8]:
            print("    %-44s ms %8.0f cells %.3e launches %5d GCUPS %.1f" % (k, v["ms"], v["cells"], v["launches"], v["cells"] / max(v["ms"], 1e-9) / 1e6))
    except Exception as e:
        print(f, "unreadable:", e)
PY
echo done
