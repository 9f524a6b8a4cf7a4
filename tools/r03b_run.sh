#!/bin/bash
# round 3, second GPU call: the fused window path (wm_window_batch) — GPU tests, then the bench with a 4096-read parity leg
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/${TAG:-r03b}
mkdir -p $OUT
echo "== window tests first =="
timeout 600 python -m pytest tests/test_window_gpu.py -m gpu -q -x > $OUT/gputest_window.txt 2>&1; echo "rc=$? $SECONDS s"; tail -15 $OUT/gputest_window.txt
echo "== all GPU tests =="
timeout 1200 python -m pytest tests -m gpu -q > $OUT/gputest.txt 2>&1; echo "rc=$? $SECONDS s"; tail -15 $OUT/gputest.txt
echo "== bench =="
( WM_BENCH_DISTINCT_BATCHES=2 WM_BENCH_CPU_SAMPLE=${CPU_SAMPLE:-4096} timeout 900 python bench.py --steps ${STEPS:-3} --warmup 1 > $OUT/bench.json 2> $OUT/bench.log ); echo "bench rc=$? $SECONDS s"; tail -4 $OUT/bench.log
python - <<'PY'
import json, os
out = os.environ["OUT"]
try:
    d = json.loads([l for l in open(out + "/bench.json").read().splitlines() if l.startswith("{")][-1])
    print("value %.4f %s ms/step %.0f parity %s" % (d["value"], d["unit"], d["ms_per_step"], d.get("parity")))
    print("host", json.dumps(d.get("host")))
    print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "gcups_all_ksw_classes")})
    for k, v in sorted(d["roofline"]["classes"].items(), key=lambda kv: -kv[1]["ms"]):
        print("  %-44s ms %9.0f cells %.3e launches %5d  GCUPS %.1f" % (k, v["ms"], v["cells"], v["launches"], v["cells"] / max(v["ms"], 1e-9) / 1e6))
except Exception as e:
    print("bench unreadable:", e)
PY
