"""On-GPU probe of the chained-workgroup ksw kernels (ksw_chain_kernel.h) against the stripe / register kernels on the same jobs: us per DP row and GCUPS
of wide hulls and long extensions, alone on the chip.   python tools/ksw_chain_probe.py [jobs scale]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from winnowmap_amd import gpu, synth

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
print("library: %s  defines: [%s]" % (gpu.LIB_PATH, gpu.build_defines()), flush=True)
TIMING = "WM_STRIPE_TIMING" in gpu.build_defines()     # diagnostic variant: per-phase cycles of the stripe / chained kernels
ctx = gpu.Context(0, 40 << 30)
sc = gpu.KswScore(2, -4, -1, 4, 2, 24, 1)
rng = np.random.default_rng(3)
SHAPES = (("p16_1500x", 1500, 256, 0x40, None), ("p16_1500a", 1500, 256, 0x08, None), ("blk_3000x", 3000, 96, 0x40, None), ("blk_3000a", 3000, 96, 0x08, None),
          ("s2_3001x_12k", 12000, 64, 0x40, 3001), ("blk2_6000x", 6000, 32, 0x40, None), ("wide_9000a", 9000, 16, 0x08, None),
          ("ext_5000x", 5000, 256, 0x40, 751), ("ext_5000a", 5000, 256, 0x08, 751), ("ext_1500x", 1500, 1024, 0x40, 751), ("ext_3000x", 3000, 512, 0x40, 751))
MODES = (("stripe/regs", 0, 2), ("chain bp2", 3, 2), ("chain bp4", 3, 4))
for name, L, njob, flag, w in SHAPES:
    njob = max(4, int(njob * scale))
    cases = []
    for it in range(njob):
        t = rng.integers(0, 4, L).astype(np.uint8)
        q = synth.mutate_codes(t, rng, 0.03, 0.03, 0.04)
        cases.append((q, t, dict(w=w if w else L + 1, zdrop=400, end_bonus=-1, flag=flag)))
    jobs, seqs = gpu.pack_jobs(cases)
    ref = None
    for mname, mode, bp in MODES:
        gpu.set_ksw_chain_routing(mode, 1, bp)
        b = ctx.ksw_prepare(sc, jobs, seqs)
        b.run()
        if TIMING:
            gpu.stripe_timing(reset=True)
        best = None
        for rep in range(2):
            t1 = time.time(); b.run(); wall = time.time() - t1
            s = b.stats()
            if best is None or s["dp_ms"] < best["dp_ms"]:
                best = dict(s, wall_ms=wall * 1e3)
        res, pool = b.fetch()
        sig = (res["score"].tolist(), res["max"].tolist(), res["n_cigar"].tolist(), int(pool.astype(np.uint64).sum()))
        if ref is None:
            ref = sig
        print("%-13s %-12s jobs=%d cells=%.3e dp=%.2f ms bt=%.2f ms -> %.1f GCUPS, %.2f us per row%s" %
              (name, mname, njob, best["cells"], best["dp_ms"], best["bt_ms"], best["cells"] / best["dp_ms"] / 1e6, best["dp_ms"] * 1e3 / (2 * L), "" if sig == ref else "   RESULTS DIFFER"), flush=True)
        if TIMING:
            t = gpu.stripe_timing(reset=True)
            rows = max(1, t["rows"])
            print("              timing: %d wavefronts, %d active wavefront-rows, %d epochs; cycles per active row: switch/prefetch %.0f  cells %.0f  wait_left %.0f  book %.0f  wait_right %.0f  publish %.0f  "
                  "epoch set-up %.0f | total per wavefront %.0f cycles, in rows %.0f %%" %
                  (t["waves"], t["rows"], t["epochs"], t["scan"] / rows, t["cells"] / rows, t["wait_left"] / rows, t["book"] / rows, t["wait_right"] / rows, t["publish"] / rows, t["epoch"] / rows,
                   t["total"] / max(1, t["waves"]), 100.0 * (t["scan"] + t["cells"] + t["wait_left"] + t["book"] + t["wait_right"] + t["publish"]) / max(1, t["total"])), flush=True)
        b.free()
gpu.set_ksw_chain_routing(1, 2048, 2)
