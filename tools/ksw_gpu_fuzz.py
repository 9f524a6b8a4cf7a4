#!/usr/bin/env python
"""GPU: the ksw kernels through the C-ABI against the oracle on random cases of every flag / band / scoring preset, under every routing:
default | every job on the chained-workgroup kernels (256- and 512-lane stripes) | two alignments per wavefront for the gap fills.
   python tools/ksw_gpu_fuzz.py [first seed] [seeds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import wmtest as W
import kswcases
from winnowmap_amd import gpu

s0, ns = (int(a) for a in (sys.argv[1:] + ["1000", "10"][len(sys.argv) - 1:])[:2])
ctx = gpu.Context(0, 6 << 30)


def run_group(cases):
    c0 = cases[0]
    sc = gpu.KswScore(c0["a"], -c0["b"], -1, c0["q_"], c0["e"], c0["q2"], c0["e2"])
    jobs, seqs = gpu.pack_jobs([(c["q"], c["t"], dict(w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])) for c in cases])
    res, pool = ctx.ksw_batch(sc, jobs, seqs)
    return [(tuple(int(res[i][k]) for k in W.EZ_FIELDS), pool[res[i]["cig_off"]:res[i]["cig_off"] + res[i]["n_cigar"]].copy()) for i in range(len(cases))]


n_al = n_bad = 0
t0 = time.time()
for seed in range(s0, s0 + ns):
    cases = kswcases.make_cases(seed, 200, max_len=[600, 1500, 3000][seed % 3], preset=seed % 5)
    for nc, sc, jobs in kswcases.dual_pairs(seed, 60, ncs=(8, 16)):
        c0 = cases[0]
        cases += [dict(j, a=c0["a"], b=c0["b"], q_=c0["q_"], e=c0["e"], q2=c0["q2"], e2=c0["e2"]) for j in jobs]
    want = []
    for c in cases:
        o = W.o_ksw_extd2(c["q"], c["t"], mat=W.simple_mat(c["a"], c["b"], 1), q=c["q_"], e=c["e"], q2=c["q2"], e2=c["e2"], w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])
        want.append((tuple(o[k] for k in W.EZ_FIELDS), o["cigar"]))
    for name, setup in (("default", lambda: (gpu.set_ksw_chain_routing(1, 2048, 2), gpu.set_ksw_dual(0))), ("all chained, 256-lane stripes", lambda: gpu.set_ksw_chain_routing(4, 2048, 2)),
                        ("all chained, 512-lane stripes", lambda: gpu.set_ksw_chain_routing(4, 2048, 4)), ("two per wavefront", lambda: (gpu.set_ksw_chain_routing(1, 2048, 2), gpu.set_ksw_dual(1)))):
        setup()
        got = run_group(cases)
        for i, (g, w_) in enumerate(zip(got, want)):
            n_al += 1
            if g[0] != w_[0] or not np.array_equal(g[1], w_[1]):
                n_bad += 1
                print("MISMATCH seed", seed, "routing", name, "case", i, "ql", len(cases[i]["q"]), "tl", len(cases[i]["t"]), "flag", hex(cases[i]["flag"]), "w", cases[i]["w"], flush=True)
gpu.set_ksw_chain_routing(1, 2048, 2); gpu.set_ksw_dual(0)
print("ksw GPU fuzz: seeds %d..%d, %d alignments under 4 routings, %d mismatches (%.0f s)" % (s0, s0 + ns - 1, n_al, n_bad, time.time() - t0))
ctx.close()
sys.exit(1 if n_bad else 0)
