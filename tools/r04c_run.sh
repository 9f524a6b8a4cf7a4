#!/bin/bash
# GPU call r04c: the epoch-structured, straight-line row loop of the stripe kernels: parity, probe, PMC instruction counts, bench A/B.
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/${TAG:-r04c}
mkdir -p $OUT
echo "== 1. stripe parity + workload test =="
timeout 200 python -m pytest tests/test_ksw_gpu.py -m gpu -x -q -k "stripe" > $OUT/gputest_stripe.txt 2>&1; rc=$?; echo "rc=$rc $SECONDS s"; tail -3 $OUT/gputest_stripe.txt
if [ $rc -ne 0 ]; then echo "stripe parity failed: stopping"; exit 1; fi
timeout 150 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -k "workload_scale or golden" > $OUT/gputest_workload.txt 2>&1; rc=$?; echo "workload rc=$rc $SECONDS s"; tail -3 $OUT/gputest_workload.txt
if [ $rc -ne 0 ]; then echo "workload test failed or hung: stopping"; exit 1; fi
echo "== 2. isolated probe (stripe classes on) =="
timeout 120 python tools/ksw_probe.py 20000 > $OUT/probe_stripe1.txt 2>&1; echo "rc=$?"; tail -8 $OUT/probe_stripe1.txt
echo "== 3. instruction counters on the probe =="
( cd /tmp && timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_sq1 -o ksw -- python $ROOT/tools/ksw_probe.py 2000 > $OUT/probe_pmc1.log 2>&1 ); echo "pmc1 rc=$?"
python - <<'PY'
import sqlite3, glob, os, collections
out = os.environ["OUT"]
for d in ("pmc_sq1",):
    for f in glob.glob(os.path.join(out, d, "**", "*.db"), recursive=True):
        db = sqlite3.connect(f)
        try:
            agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
            for k, c, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
                agg[k][c] += v
            with open(os.path.join(out, d + "_summary.txt"), "w") as fo:
                for k in sorted(agg):
                    if "ksw" in k:
                        fo.write(k[:100] + "\n")
                        for c in sorted(agg[k]):
                            fo.write("    %-28s %.6g\n" % (c, agg[k][c]))
        except Exception as e:
            print(d, "summary failed:", e)
        os.remove(f)
PY
grep -A8 "stripe" $OUT/pmc_sq1_summary.txt | head -60
echo "== 4. bench A/B (16384 reads per step, 4 steps) =="
run_bench() { # tag, env...
  local tag=$1; shift
  local t0=$SECONDS
  ( env "$@" WM_BENCH_CPU_SAMPLE=0 timeout 150 python bench.py --steps 4 --warmup 2 --reads-per-step 16384 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $((SECONDS-t0))s $(cut -c1-100 $OUT/bench_$tag.json)"
}
run_bench stripe1
run_bench stripe0 WM_KSW_STRIPE=0
run_bench stripe1_r8_1024 WM_KSW_STRIPE_ROWS8=1024
echo "== summary ($SECONDS s) =="
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ["OUT"], "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-28s %.4f %s  ms/step %.0f cpu_s %.1f" % (os.path.basename(f), d["value"], d["unit"], d["ms_per_step"], d["host"]["process_cpu_s"]))
        for k, v in sorted(d["roofline"]["classes"].items(), key=lambda kv: -kv[1]["ms"])[:12]:
            print("    %-44s ms %8.0f cells %.3e launches %5d GCUPS %.1f" % (k, v["ms"], v["cells"], v["launches"], v["cells"] / max(v["ms"], 1e-9) / 1e6))
    except Exception as e:
        print(f, "unreadable:", e)
PY
