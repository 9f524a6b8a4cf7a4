"""One shape of tools/ksw_chain_probe.py on one routing, a few runs: the workload of the --pmc passes of tools/r06h_run.sh.
   python tools/ksw_one_shape.py <L> <jobs> <flag hex> <w or 0> <mode> <bp>"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from winnowmap_amd import gpu, synth
L, njob, flag, w, mode, bp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3], 16), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
ctx = gpu.Context(0, 24 << 30)
sc = gpu.KswScore(2, -4, -1, 4, 2, 24, 1)
rng = np.random.default_rng(3)
cases = []
for it in range(njob):
    t = rng.integers(0, 4, L).astype(np.uint8)
    cases.append((synth.mutate_codes(t, rng, 0.03, 0.03, 0.04), t, dict(w=w if w else L + 1, zdrop=400, end_bonus=-1, flag=flag)))
jobs, seqs = gpu.pack_jobs(cases)
gpu.set_ksw_chain_routing(mode, 1, bp)
b = ctx.ksw_prepare(sc, jobs, seqs)
for rep in range(3):
    b.run()
s = b.stats()
print("L %d jobs %d flag 0x%x w %d mode %d bp %d: cells %.3e dp %.2f ms rows %d" % (L, njob, flag, w, mode, bp, s["cells"], s["dp_ms"], 2 * L - 1), flush=True)
