"""Quick on-GPU probe of the ksw kernels: GCUPS and traceback GB/s on map-ont shaped segment batches."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from winnowmap_amd import gpu
import kswcases

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
TIMING = "WM_STRIPE_TIMING" in gpu.build_defines()     # diagnostic variant: per-phase cycles of the stripe-pipelined kernel (ksw_stripe_kernel.h)
NOSTORE = "WM_KSW_NOSTORE" in gpu.build_defines()      # timing-only variant: the traceback is never written, the backtrack may fail on what it finds


def run(b):
    try:
        b.run()
    except gpu.WmError:
        if not NOSTORE:
            raise


print("library: %s  defines: [%s]" % (gpu.LIB_PATH, gpu.build_defines()), flush=True)
ctx = gpu.Context(0, 24 << 30)
sc = gpu.KswScore(2, -4, -1, 4, 2, 24, 1)
# ...x = the same shapes as extensions: exact maximum + z-drop (KSW_EZ_EXTZ_ONLY), the EXACT kernel instantiations
for name, mean, w in (("ont300", 300, 751), ("ont150", 150, 751), ("ont600", 600, 751), ("ont600clip", 600, 500), ("ont300x", 300, 751), ("ont600x", 600, 751), ("ont600clipx", 600, 300)):
    t0 = time.time()
    base = kswcases.ont_segments(1, 2000, mean=mean, w=w)
    if name.endswith("x"):
        for c in base:
            c["flag"] = 0x40
    cases = [base[i % len(base)] for i in range(n)]
    jobs, seqs = gpu.pack_jobs([(c["q"], c["t"], dict(w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])) for c in cases])
    tgen = time.time() - t0
    b = ctx.ksw_prepare(sc, jobs, seqs)
    run(b)
    best = None
    for rep in range(3):
        t1 = time.time(); run(b); wall = time.time() - t1
        s = b.stats()
        if best is None or s["dp_ms"] < best["dp_ms"]:
            best = dict(s, wall_ms=wall * 1e3)
    print("%-11s jobs=%d cells=%.3e dp=%.2f ms bt=%.2f ms wall=%.2f ms  -> %.1f GCUPS (dp), tb %.1f GB/s, gen %.1fs" %
          (name, n, best["cells"], best["dp_ms"], best["bt_ms"], best["wall_ms"], best["cells"] / best["dp_ms"] / 1e6,
           best["tb_bytes"] / best["dp_ms"] / 1e6, tgen), flush=True)
    b.free()
# wide hulls: the multi-wave classes alone on the chip (ksw_pmulti_kernel<4,4> / <4,8> / <8,8>) — their rate here against their rate inside the
# mapper (bench.py `classes`) separates what the kernel costs from what sharing the chip costs
from winnowmap_amd import synth
rng = np.random.default_rng(3)
# ...ext_5000x: extensions inside the 751 band (8-pair class by pitch; long enough for the stripe kernel <2,4> by default routing)
for name, L, njob, flag in (("p16_1500x", 1500, 256, 0x40), ("p16_1500a", 1500, 256, 0x08), ("blk_3000x", 3000, 96, 0x40), ("blk_3000a", 3000, 96, 0x08), ("blk2_6000x", 6000, 32, 0x40),
                            ("ext_5000x", 5000, 256, 0x40), ("ext_5000a", 5000, 256, 0x08), ("ext_1500x", 1500, 1024, 0x40)):
    cases = []
    for it in range(njob):
        t = rng.integers(0, 4, L).astype(np.uint8)
        q = synth.mutate_codes(t, rng, 0.03, 0.03, 0.04)
        cases.append((q, t, dict(w=751 if name.startswith("ext") else L + 1, zdrop=400, end_bonus=-1, flag=flag)))
    jobs, seqs = gpu.pack_jobs(cases)
    b = ctx.ksw_prepare(sc, jobs, seqs)
    run(b)
    if TIMING:
        gpu.stripe_timing(reset=True)
    best = None
    for rep in range(2):
        t1 = time.time(); run(b); wall = time.time() - t1
        s = b.stats()
        if best is None or s["dp_ms"] < best["dp_ms"]:
            best = dict(s, wall_ms=wall * 1e3)
    print("%-11s jobs=%d cells=%.3e dp=%.2f ms bt=%.2f ms wall=%.2f ms  -> %.1f GCUPS (dp), %.2f us per row" %
          (name, njob, best["cells"], best["dp_ms"], best["bt_ms"], best["wall_ms"], best["cells"] / best["dp_ms"] / 1e6, best["dp_ms"] * 1e3 / (2 * L)), flush=True)
    if TIMING:          # where a stripe wavefront's cycles go (WM_STRIPE_TIMING build): per ACTIVE row of a wavefront, and as shares of its time in the kernel
        t = gpu.stripe_timing(reset=True)
        rows = max(1, t["rows"])
        print("            stripe timing: %d wavefronts, %d active wavefront-rows (%.1f per job-row), %d epochs; cycles per active row: cells %.0f  wait_left %.0f  book %.0f  "
              "wait_right %.0f  publish %.0f  epoch set-up %.0f  scan/idle %.0f | share of kernel time: cells %.0f %%  waits %.0f %%  book %.0f %%  publish %.0f %%  scan/idle %.0f %%" %
              (t["waves"], t["rows"], t["rows"] / (2.0 * 2 * L * njob), t["epochs"], t["cells"] / rows, t["wait_left"] / rows, t["book"] / rows, t["wait_right"] / rows, t["publish"] / rows,
               t["epoch"] / rows, t["scan"] / rows, 100.0 * t["cells"] / max(1, t["total"]), 100.0 * (t["wait_left"] + t["wait_right"]) / max(1, t["total"]),
               100.0 * t["book"] / max(1, t["total"]), 100.0 * t["publish"] / max(1, t["total"]), 100.0 * t["scan"] / max(1, t["total"])), flush=True)
    b.free()
