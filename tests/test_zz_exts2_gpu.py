"""GPU parity of the splice-aware extension: wm_ksw_exts2_batch (ksw_exts2_kernel.h through the C-ABI) vs the oracle's restatement of
ksw_exts2_sse, which tests/test_oracle_vs_ref.py pins to the reference's own function. The kernel is bit-exact on the wavefront emulator
(tests/test_kernels_emu.py) and on the GPU (first run: round 3, profiles/r03a_first_run.txt)."""
import os
import numpy as np
import pytest
import wmtest as W
import kswcases
from winnowmap_amd import gpu

pytestmark = [pytest.mark.gpu]


@pytest.mark.parametrize("with_junc", [False, True])
def test_exts2_batches_match_oracle(with_junc):
    ctx = gpu.Context(0, 4 << 30)
    try:
        cases = kswcases.make_splice_cases(41, 480) + kswcases.make_splice_cases(42, 60, max_exon=300, max_intron=2500)
        groups = {}
        for c in cases:          # one scoring set and one pair of splice parameters per batch
            groups.setdefault((c["a"], c["b"], c["q_"], c["e"], c["q2"], c["noncan"], c["junc_bonus"]), []).append(c)
        n_intron = 0
        for (a, b, q, e, q2, noncan, jb), cs in groups.items():
            sc = gpu.KswScore(a, -b, -1, q, e, q2, 0)
            jobs, seqs = gpu.pack_jobs([(c["q"], c["t"], dict(w=-1, zdrop=c["zdrop"], end_bonus=0, flag=c["flag"])) for c in cs])
            junc = None
            if with_junc:
                junc = np.zeros(len(seqs), np.uint8)
                for j, c in zip(jobs, cs):
                    if c["junc"] is not None:
                        junc[j["t_off"]:j["t_off"] + j["tlen"]] = c["junc"]
            res, pool = ctx.ksw_exts2_batch(sc, noncan, jb, jobs, seqs, junc)
            for i, c in enumerate(cs):
                o = W.o_ksw_exts2(c["q"], c["t"], mat=W.simple_mat(a, b, 1), q=q, e=e, q2=q2, noncan=noncan, zdrop=c["zdrop"], junc_bonus=jb, flag=c["flag"],
                                  junc=c["junc"] if with_junc else None)
                g = res[i]
                cig = pool[g["cig_off"]:g["cig_off"] + g["n_cigar"]]
                assert all(int(g[k]) == o[k] for k in W.EZ_FIELDS), (i, hex(c["flag"]), {k: (int(g[k]), o[k]) for k in W.EZ_FIELDS if int(g[k]) != o[k]})
                assert np.array_equal(cig, o["cigar"]), (i, hex(c["flag"]), W.cigar_str(cig)[:60], W.cigar_str(o["cigar"])[:60])
                n_intron += any((int(x) & 0xf) == 3 for x in o["cigar"])
        assert n_intron > 150
        # argument checks follow the reference's early returns (src/ksw2_exts2_sse.c:66, :84)
        with pytest.raises(Exception):
            ctx.ksw_exts2_batch(gpu.KswScore(1, -2, -1, 2, 1, 3, 0), 9, 9, jobs[:1], seqs)
    finally:
        ctx.close()
