#!/bin/bash
# round 3, eleventh GPU call: (1) the isolated ksw probe with three builds — default (loads_land), round-2 wait placement (WM_LOADS_LAND=0),
# timing-only build without traceback stores; (2) bench A/B at identical settings (4 steps, 2 warm-up, two mini-batches in flight):
# default vs WM_LOADS_LAND=0, and the default build with 8 / 12 host threads per mapping call (two calls run at once: 16 + 16 threads
# hit the container's 16-CPU quota, profiles/r03j_bench.json cpu_quota_throttled)
set -u
export TMPDIR=/tmp
ROOT=$PWD
export OUT=$ROOT/gpurun_out/${TAG:-r03k}
mkdir -p $OUT
echo "== probe: default / round-2 wait placement / no stores =="
for v in "" _ll0 _nostore; do
  ( WM_LIBWMGPU=$ROOT/winnowmap_amd/libwmgpu$v.so timeout 240 python tools/ksw_probe.py 20000 > $OUT/ksw_probe$v.txt 2>&1 ); echo "probe$v rc=$?"; head -9 $OUT/ksw_probe$v.txt
done
run_bench() { # tag, args..., env via WM_ENV
  local tag=$1; shift
  local t0=$SECONDS
  ( env ${WM_ENV:-WM_X=1} timeout 900 python bench.py "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log ); echo "[$tag] rc=$? $((SECONDS-t0))s $(tail -1 $OUT/bench_$tag.log | cut -c1-200)"
}
export WM_BENCH_DISTINCT_BATCHES=2 WM_BENCH_CPU_SAMPLE=0
WM_ENV="WM_X=1" run_bench new --steps 4 --warmup 2
WM_ENV="WM_LIBWMGPU=$ROOT/winnowmap_amd/libwmgpu_ll0.so" run_bench ll0 --steps 4 --warmup 2
WM_ENV="WM_X=1" run_bench new_t8 --steps 4 --warmup 2 --threads 8
WM_ENV="WM_X=1" run_bench new_t12 --steps 4 --warmup 2 --threads 12
python - <<'PY'
import os, json, glob
out = os.environ["OUT"]
for f in sorted(glob.glob(os.path.join(out, "bench_*.json"))):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        h = d["host"]
        print("%-22s %.4f Gbp/s ms/step %5.0f cpu/step %.1f util %.2f thr %s | calls w %d k %d | throttled %s | defines [%s]" % (
              os.path.basename(f), d["value"], d["ms_per_step"], h["process_cpu_s"] / d["steps"], h["cpu_utilisation"], h.get("host_threads"), h["batched_calls"]["window"], h["batched_calls"]["ksw"],
              h.get("cpu_quota_throttled"), d["config"].get("variants", {}).get("kernel_defines")))
        for k, v in sorted(d["roofline"]["classes"].items(), key=lambda kv: -kv[1]["ms"])[:6]:
            print("    %-44s ms %8.0f cells %.3e launches %5d GCUPS %.1f" % (k, v["ms"], v["cells"], v["launches"], v["cells"] / max(v["ms"], 1e-9) / 1e6))
    except Exception as e:
        print(f, "unreadable:", e)
PY
echo done
