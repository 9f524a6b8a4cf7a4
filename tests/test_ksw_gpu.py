"""GPU parity: the HIP ksw kernels (through the C-ABI) vs the oracle, bit-exact (integer DP)."""
import numpy as np
import pytest
import wmtest as W
import kswcases
from winnowmap_amd import gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = gpu.Context(0, 4 << 30)
    yield c
    c.close()


def _run_group(ctx, cases):
    """cases share one scoring preset (the C-ABI takes one score set per batch)."""
    c0 = cases[0]
    sc = gpu.KswScore(c0["a"], -c0["b"], -1, c0["q_"], c0["e"], c0["q2"], c0["e2"])
    jobs, seqs = gpu.pack_jobs([(c["q"], c["t"], dict(w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])) for c in cases])
    res, pool = ctx.ksw_batch(sc, jobs, seqs)
    bad = []
    for i, c in enumerate(cases):
        o = W.o_ksw_extd2(c["q"], c["t"], mat=W.simple_mat(c["a"], c["b"], 1), q=c["q_"], e=c["e"], q2=c["q2"], e2=c["e2"],
                          w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])
        g = res[i]
        cig = pool[g["cig_off"]:g["cig_off"] + g["n_cigar"]]
        ok = all(int(g[k]) == o[k] for k in W.EZ_FIELDS) and np.array_equal(cig, o["cigar"])
        if not ok:
            bad.append((i, {k: (int(g[k]), o[k]) for k in W.EZ_FIELDS if int(g[k]) != o[k]}, W.cigar_str(cig)[:50], W.cigar_str(o["cigar"])[:50]))
    return bad


@pytest.mark.parametrize("preset", [0, 1, 2, 3, 4])
def test_random_cases_all_flags(ctx, preset):
    cases = kswcases.make_cases(100 + preset, 240, max_len=900, preset=preset)
    bad = _run_group(ctx, cases)
    assert not bad, bad[:3]


def test_ont_shaped_batch(ctx):
    cases = kswcases.ont_segments(7, 600)
    bad = _run_group(ctx, cases)
    assert not bad, bad[:3]


def test_band_clipped_and_stale_lanes(ctx):
    # long segments with narrow bands: out-of-band (stale) lanes feed back and int8 values wrap in the reference
    rng = np.random.default_rng(3)
    cases = []
    for it in range(60):
        c = kswcases.make_cases(500 + it, 1, max_len=950, preset=0)[0]
        c["w"] = [5, 10, 33, 50, 120, 200][it % 6]
        cases.append(c)
    bad = _run_group(ctx, cases)
    assert not bad, bad[:3]


def test_degenerate_and_tiny(ctx):
    sc = gpu.KswScore(2, -4, -1, 4, 2, 24, 1)
    one = np.array([2], np.uint8)
    pairs = [(one, one), (one, np.array([1, 2, 3], np.uint8)), (np.array([0, 1, 2, 3] * 5, np.uint8), one)]
    jobs, seqs = gpu.pack_jobs(pairs, flag=0x40)
    res, pool = ctx.ksw_batch(sc, jobs, seqs)
    for i, (q, t) in enumerate(pairs):
        o = W.o_ksw_extd2(q, t, flag=0x40)
        assert all(int(res[i][k]) == o[k] for k in W.EZ_FIELDS)
        assert np.array_equal(pool[res[i]["cig_off"]:res[i]["cig_off"] + res[i]["n_cigar"]], o["cigar"])
    # empty batch
    res, pool = ctx.ksw_batch(sc, np.zeros(0, gpu.KSW_JOB_DTYPE), np.zeros(1, np.uint8))
    assert len(res) == 0 and len(pool) == 0


def test_round_trip_property_full_size(ctx):
    # size-independent property at workload scale: CIGAR consumes exactly the aligned prefixes and re-scoring
    # the CIGAR reproduces ez.score for global (gap-fill) jobs
    cases = kswcases.ont_segments(11, 3000)
    sc = gpu.KswScore(2, -4, -1, 4, 2, 24, 1)
    jobs, seqs = gpu.pack_jobs([(c["q"], c["t"], dict(w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=0x08)) for c in cases])
    res, pool = ctx.ksw_batch(sc, jobs, seqs)
    for i, c in enumerate(cases):
        cig = pool[res[i]["cig_off"]:res[i]["cig_off"] + res[i]["n_cigar"]]
        qi = ti = 0
        score = 0
        for op in cig:
            l, o = int(op) >> 4, int(op) & 0xf
            if o == 0:
                eq = c["q"][qi:qi + l] == c["t"][ti:ti + l]
                score += int(eq.sum()) * 2 - int((~eq).sum()) * 4
                qi += l; ti += l
            elif o == 1:
                score -= min(4 + 2 * l, 24 + l); qi += l
            else:
                score -= min(4 + 2 * l, 24 + l); ti += l
        assert qi == len(c["q"]) and ti == len(c["t"])
        assert score == int(res[i]["score"]), (i, score, int(res[i]["score"]))


def test_wide_band_jobs_block_and_generic_kernels(ctx):
    # hulls wider than the register window: stage-2 gap fills (w = 3001 -> multi-wave LDS kernel) and one job whose
    # hull exceeds even that (global-scratch generic kernel)
    from winnowmap_amd import synth
    rng = np.random.default_rng(21)
    cases = []
    for it in range(14):
        tl = int(rng.integers(1100, 2800))
        t = rng.integers(0, 4, tl).astype(np.uint8)
        q = synth.mutate_codes(t, rng, 0.04, 0.04, 0.05) if it % 4 else rng.integers(0, 4, int(rng.integers(1100, 2200))).astype(np.uint8)
        if it % 5 == 1:
            q[int(rng.integers(0, len(q)))] = 4
        cases.append(dict(q=q, t=t, a=2, b=4, q_=4, e=2, q2=24, e2=1, w=[3001, 1500, 3001, 2000][it % 4], zdrop=[400, 200, -1][it % 3],
                          end_bonus=-1, flag=kswcases.FLAGS[it % 6]))
    for tl, fl in ((3400, 0x08), (7600, 0x08), (7300, 0x40)):       # unbanded: LDS kernel with the 8192-lane window, then the global-state kernel
        t = rng.integers(0, 4, tl).astype(np.uint8)
        cases.append(dict(q=synth.mutate_codes(t, rng, 0.03, 0.03, 0.03), t=t, a=2, b=4, q_=4, e=2, q2=24, e2=1, w=-1, zdrop=400, end_bonus=-1, flag=fl))
    bad = _run_group(ctx, cases)
    assert not bad, bad[:3]
