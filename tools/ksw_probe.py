"""Quick on-GPU probe of the ksw kernels: GCUPS and traceback GB/s on map-ont shaped segment batches."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from winnowmap_amd import gpu
import kswcases

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
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
    b.run()
    best = None
    for rep in range(3):
        t1 = time.time(); b.run(); wall = time.time() - t1
        s = b.stats()
        if best is None or s["dp_ms"] < best["dp_ms"]:
            best = dict(s, wall_ms=wall * 1e3)
    print("%-11s jobs=%d cells=%.3e dp=%.2f ms bt=%.2f ms wall=%.2f ms  -> %.1f GCUPS (dp), tb %.1f GB/s, gen %.1fs" %
          (name, n, best["cells"], best["dp_ms"], best["bt_ms"], best["wall_ms"], best["cells"] / best["dp_ms"] / 1e6,
           best["tb_bytes"] / best["dp_ms"] / 1e6, tgen), flush=True)
    b.free()
