"""Long fuzz of the emulated ksw kernels against the oracle (the shapes of tests/test_kernels_emu.py, more seeds).
  python tools/emu_fuzz.py [first_seed] [n_seeds] [variant: default|readlane]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import wmtest as W
import kswcases
from test_kernels_emu import _load_emu, emu_ksw, KSW_VARIANTS

s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 20
var = sys.argv[3] if len(sys.argv) > 3 else "default"
E = _load_emu(KSW_VARIANTS[var])
runs = bad = 0
for seed in range(s0, s0 + ns):
    for c in kswcases.make_cases(seed, 90, max_len=[300, 600, 1200][seed % 3]):
        o = W.o_ksw_extd2(c["q"], c["t"], mat=W.simple_mat(c["a"], c["b"], 1), q=c["q_"], e=c["e"], q2=c["q2"], e2=c["e2"],
                          w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])
        for force in (-1, 3, 11, 19, 2, 10, 18, 0, 100, 102, 103, 220, 222, 230, 232, 243):
            n, ez, cig, klass = emu_ksw(E, c, force)
            if n < 0:
                continue
            runs += 1
            if [int(x) for x in ez] != [o[k] for k in W.EZ_FIELDS] or not np.array_equal(cig, o["cigar"]):
                bad += 1
                print("MISMATCH seed", seed, "force", force, "klass", klass, "flag", hex(c["flag"]), "w", c["w"], "ql", len(c["q"]), "tl", len(c["t"]), flush=True)
print("emu fuzz (%s): seeds %d..%d, %d runs, %d mismatches" % (var, s0, s0 + ns - 1, runs, bad))
