#!/usr/bin/env python
"""ksw_dp_dual (two alignments per wavefront, winnowmap_amd/csrc/ksw_dual_kernel.h) on the wavefront emulator against the oracle, for as many seeds as asked:
   python tools/dual_fuzz.py [first seed] [seeds] [pairs per seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import wmtest as W
import kswcases
import test_kernels_emu as T
E = T._load_emu()
s0, ns, per = (int(a) for a in (sys.argv[1:] + ["1", "20", "300"][len(sys.argv) - 1:])[:3])
n_al, t0 = 0, time.time()
for seed in range(s0, s0 + ns):
    for nc, sc, jobs in kswcases.dual_pairs(seed, per):
        rc, out = T.emu_ksw_dual(E, nc, sc, jobs)
        if rc == -1:
            continue
        assert rc == 0, (seed, rc)
        for j, (ez, cig) in zip(jobs, out):
            o = W.o_ksw_extd2(j["q"], j["t"], mat=W.simple_mat(sc["a"], sc["b"], 1), q=sc["q_"], e=sc["e"], q2=sc["q2"], e2=sc["e2"], w=j["w"], zdrop=-1, end_bonus=0, flag=j["flag"])
            assert [int(v) for v in ez] == [o[k] for k in W.EZ_FIELDS] and np.array_equal(cig, o["cigar"]), (seed, nc, len(j["q"]), len(j["t"]), hex(j["flag"]), j["w"])
            n_al += 1
print("seeds %d..%d: %d alignments, 0 mismatches (%.0f s)" % (s0, s0 + ns - 1, n_al, time.time() - t0))
