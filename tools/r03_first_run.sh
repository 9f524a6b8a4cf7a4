#!/bin/bash
# The first GPU call of round 3 — validates and times what round 2 prepared after its GPU minutes were spent:
#   gpurun --timeout 3300 -- 'bash tools/r03_first_run.sh'
# 1) the whole GPU suite incl. the opt-in tests (packed multi-wave kernel, splice kernel);
# 2) bench.py (config 2) with the library as built: the restructured single-wave kernels + -disable-promote-alloca-to-vector;
# 3) the same with WM_KSW_PMULTI=1 (BLOCK / BLOCK2 classes on ksw_dp_pmulti) and =2 (the 16-pair classes as well);
# 4) rebuild with WM_KERNEL_DEFINES=WM_KSW_ROR=1 (wave_ror + v_perm neighbours), ksw GPU tests, bench;
# 5) isolated ksw probe (GCUPS per size) for the default build.
# Everything lands in gpurun_out/r03a/. CPU baseline legs are skipped after the first bench (WM_BENCH_CPU_SAMPLE=0).
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/r03a
mkdir -p $OUT
STEPS=${STEPS:-2}
run_bench() { # tag, env...
  local tag=$1; shift
  ( env "$@" python bench.py --steps $STEPS --warmup 1 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $(cut -c1-400 $OUT/bench_$tag.json)"
}
echo "== 1. GPU tests incl. opt-in =="
WM_TEST_OPTIN=1 timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/gputest.txt 2>&1; echo "rc=$?"; tail -5 $OUT/gputest.txt
echo "== 2. bench, library as built =="
run_bench default WM_DUMMY=1
echo "== 3. packed multi-wave kernel =="
run_bench pmulti1 WM_KSW_PMULTI=1 WM_BENCH_CPU_SAMPLE=0
run_bench pmulti2 WM_KSW_PMULTI=2 WM_BENCH_CPU_SAMPLE=0
run_bench coopbt WM_KSW_COOP_BT=1 WM_BENCH_CPU_SAMPLE=0
echo "== 5. ksw probe (default build) =="
timeout 600 python tools/ksw_probe.py > $OUT/ksw_probe_default.txt 2>&1; tail -8 $OUT/ksw_probe_default.txt
echo "== 4. WM_KSW_ROR build =="
WM_KERNEL_DEFINES="WM_KSW_ROR=1" python -c "from winnowmap_amd import build; build.build_gpu(force=True, verbose=True)" > $OUT/build_ror.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests/test_ksw_gpu.py tests/test_e2e_gpu.py -m gpu -q -x > $OUT/gputest_ror.txt 2>&1; echo "rc=$?"; tail -3 $OUT/gputest_ror.txt
run_bench ror WM_BENCH_CPU_SAMPLE=0
run_bench ror_pmulti2 WM_KSW_PMULTI=2 WM_BENCH_CPU_SAMPLE=0
timeout 600 python tools/ksw_probe.py > $OUT/ksw_probe_ror.txt 2>&1; tail -8 $OUT/ksw_probe_ror.txt
python -c "from winnowmap_amd import build; build.build_gpu(force=True)" > /dev/null 2>&1     # back to the default build
echo "== summary =="
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ.get("OUT", "gpurun_out/r03a"), "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-28s %.4f %s  ms/step %.0f  parity %s  roofline %s" % (os.path.basename(f), d["value"], d["unit"], d["ms_per_step"], d.get("parity"), {k: d["roofline"][k] for k in ("kernel", "achieved", "frac") if k in d.get("roofline", {})}))
    except Exception as e:
        print(f, "unreadable:", e)
PY
