#!/bin/bash
# The first GPU call of round 3 — validates and times what round 2 prepared after its GPU minutes were spent:
#   gpurun --timeout 2400 -- 'bash tools/r03_first_run.sh'
# 1) the whole GPU suite incl. the opt-in tests (packed multi-wave kernel, splice kernel, wave backtrack);
# 2) bench.py (config 2) with the library as built; 3) WM_KSW_PMULTI=1|2, WM_KSW_COOP_BT=1;
# 4) isolated ksw probe (GCUPS per size) + SQ counter passes on it; 5) one RCCL-path run on one GPU;
# 6) rebuild with WM_KERNEL_DEFINES=WM_KSW_ROR=1: ksw tests, bench, probe.
# Everything lands in gpurun_out/r03a/.
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/r03a
mkdir -p $OUT
STEPS=${STEPS:-2}
export WM_BENCH_DISTINCT_BATCHES=2
run_bench() { # tag, env...
  local tag=$1; shift
  local t0=$SECONDS
  ( env "$@" timeout 600 python bench.py --steps $STEPS --warmup 1 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $((SECONDS-t0))s $(cut -c1-200 $OUT/bench_$tag.json)"
}
echo "== 1. GPU tests incl. opt-in =="
WM_TEST_OPTIN=1 timeout 1200 python -m pytest tests -m gpu -q > $OUT/gputest.txt 2>&1; echo "rc=$? $SECONDS s"; tail -15 $OUT/gputest.txt
echo "== 2. bench, library as built =="
run_bench default WM_BENCH_CPU_SAMPLE=0
echo "== 3. variants =="
run_bench pmulti1 WM_KSW_PMULTI=1 WM_BENCH_CPU_SAMPLE=0
run_bench pmulti2 WM_KSW_PMULTI=2 WM_BENCH_CPU_SAMPLE=0
run_bench coopbt WM_KSW_COOP_BT=1 WM_BENCH_CPU_SAMPLE=0
echo "== 4. ksw probe (default build) + SQ counters =="
timeout 300 python tools/ksw_probe.py > $OUT/ksw_probe_default.txt 2>&1; tail -8 $OUT/ksw_probe_default.txt
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_sq1 -o ksw -- python $ROOT/tools/ksw_probe.py 20000 > $OUT/probe_pmc1.log 2>&1 ); echo "pmc1 rc=$?"
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU -d $OUT/pmc_sq2 -o ksw -- python $ROOT/tools/ksw_probe.py 20000 > $OUT/probe_pmc2.log 2>&1 ); echo "pmc2 rc=$?"
python - <<'PY'
import sqlite3, glob, os, collections
out = os.environ["OUT"]
for d in ("pmc_sq1", "pmc_sq2"):
    for f in glob.glob(os.path.join(out, d, "**", "*.db"), recursive=True):
        db = sqlite3.connect(f)
        try:
            agg = collections.defaultdict(lambda: collections.defaultdict(float))
            for k, c, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
                agg[k][c] += v
            with open(os.path.join(out, d + "_summary.txt"), "w") as fo:
                for k in sorted(agg):
                    if "ksw" in k:
                        fo.write(k[:120] + "\n")
                        for c in sorted(agg[k]):
                            fo.write("    %-28s %.6g\n" % (c, agg[k][c]))
        except Exception as e:
            print(d, "summary failed:", e)
        os.remove(f)      # the raw db is large; keep the summary
PY
cat $OUT/pmc_sq1_summary.txt 2>/dev/null | head -60
echo "== 5. RCCL path on one GPU =="
( WM_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 WM_BENCH_CPU_SAMPLE=0 WM_BENCH_READS=8192 timeout 600 python bench.py --gpus 1 --steps 1 --warmup 0 > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.log ); echo "forcedist rc=$? $(cut -c1-200 $OUT/bench_forcedist.json)"; tail -3 $OUT/bench_forcedist.log
echo "== 6. WM_KSW_ROR build =="
WM_KERNEL_DEFINES="WM_KSW_ROR=1" python -c "from winnowmap_amd import build; build.build_gpu(force=True, verbose=True)" > $OUT/build_ror.log 2>&1; echo "build rc=$? $SECONDS s"
timeout 600 python -m pytest tests/test_ksw_gpu.py tests/test_e2e_gpu.py -m gpu -q > $OUT/gputest_ror.txt 2>&1; echo "rc=$?"; tail -3 $OUT/gputest_ror.txt
run_bench ror WM_BENCH_CPU_SAMPLE=0
timeout 300 python tools/ksw_probe.py > $OUT/ksw_probe_ror.txt 2>&1; tail -8 $OUT/ksw_probe_ror.txt
echo "== summary ($SECONDS s) =="
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ.get("OUT", "gpurun_out/r03a"), "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-28s %.4f %s  ms/step %.0f  roofline %s" % (os.path.basename(f), d["value"], d["unit"], d["ms_per_step"], {k: d["roofline"][k] for k in ("kernel", "achieved", "frac") if k in d.get("roofline", {})}))
    except Exception as e:
        print(f, "unreadable:", e)
PY
