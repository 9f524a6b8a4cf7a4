"""GPU parity: the HIP ksw kernels (through the C-ABI) vs the oracle, bit-exact (integer DP)."""
import os
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


def _default_chain_routing():
    gpu.set_ksw_chain_routing(int(os.environ.get("WM_KSW_CHAIN", 1)) & 7, int(os.environ.get("WM_KSW_CHAIN_ROWS", 2048)), 4 if os.environ.get("WM_KSW_CHAIN_BP") == "4" else 2)


@pytest.fixture(autouse=True)
def chain_routing_off_unless_asked():
    """the chained-workgroup kernels (round 6) take every wide or long job by default; the tests of the older kernels below switch them off so that those
    kernels still run, the chain tests at the end of the file choose their own mode"""
    gpu.set_ksw_chain_routing(0, -1, -1)
    yield
    _default_chain_routing()


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


def test_band_sitting_on_the_last_target_lane(ctx):
    # target lengths of 16k + 1 with narrow bands: en0 == st == tlen - 1 on rows that do not re-base, so lane en0 takes H from the lane below the
    # window as the LAST re-base left it (a bug of the first multi-wave packed kernel, found on the emulator)
    from winnowmap_amd import synth
    rng = np.random.default_rng(5)
    cases = []
    for it, (ql, tl) in enumerate(((214, 209), (489, 193), (150, 129), (333, 321), (90, 65), (700, 641), (1500, 1297), (260, 257))):
        t = rng.integers(0, 4, tl).astype(np.uint8)
        q = np.concatenate([synth.mutate_codes(t, rng, 0.03, 0.02, 0.02), rng.integers(0, 4, ql).astype(np.uint8)])[:ql]
        cases.append(dict(q=q, t=t, a=2, b=4, q_=4, e=2, q2=24, e2=1, w=[5, 5, 3, 9, 5, 17, 33, 2][it], zdrop=[200, 400, -1, 100, 50, 400, 200, -1][it],
                          end_bonus=[0, -1, 10][it % 3], flag=[0x80, 0, 0x40, 0x80, 0, 0x42, 0x00, 0xC0][it]))
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


def test_wide_band_jobs_packed_multiwave_kernel(ctx, monkeypatch):
    """(first run on a GPU: round 3, profiles/r03a_first_run.txt; WM_KSW_PMULTI=2 is the library default since) WM_KSW_PMULTI=1 routes the BLOCK / BLOCK2 classes to ksw_pmulti_kernel<4,8> / <8,8> (ksw_packed_multi_kernel.h): same cases, same bar."""
    monkeypatch.setenv("WM_KSW_PMULTI", "1")
    test_wide_band_jobs_block_and_generic_kernels(ctx)
    # 2: the 16-pair register classes (hulls of 1009..2032 lanes) run on ksw_pmulti_kernel<4,4> as well
    monkeypatch.setenv("WM_KSW_PMULTI", "2")
    test_wide_band_jobs_block_and_generic_kernels(ctx)
    rng = np.random.default_rng(8)
    from winnowmap_amd import synth
    cases = []
    for it in range(24):
        tl = int(rng.integers(1050, 2000))
        t = rng.integers(0, 4, tl).astype(np.uint8)
        q = synth.mutate_codes(t, rng, 0.04, 0.04, 0.05)
        if it % 6 == 2:
            t[int(rng.integers(0, tl))] = 4
        cases.append(dict(q=q, t=t, a=2, b=4, q_=4, e=2, q2=24, e2=1, w=[-1, 2500, 1100, 700][it % 4], zdrop=[400, 200, -1][it % 3],
                          end_bonus=[-1, 10][it % 2], flag=kswcases.FLAGS[it % 6]))
    bad = _run_group(ctx, cases)
    assert not bad, bad[:3]


def test_wave_cooperative_backtrack(ctx, monkeypatch):
    """WM_KSW_COOP_BT=1: one wavefront per alignment walks the traceback through LDS tiles (ksw_backtrack_wave); same cases, same bar."""
    monkeypatch.setenv("WM_KSW_COOP_BT", "1")
    test_random_cases_all_flags(ctx, 0)
    test_band_clipped_and_stale_lanes(ctx)
    test_wide_band_jobs_block_and_generic_kernels(ctx)
    test_degenerate_and_tiny(ctx)


def test_position_jobs_equal_byte_jobs(tmp_path):
    """wm_ksw_batch_pos expands its operands inside HBM from the resident read codes (two-strand space, src/align.c:871-877) and the
    packed reference (mm_idx_getseq, src/index.c:161-171): results must equal wm_ksw_batch on the same operands as bytes, which the
    other tests pin to the oracle. Covers both strands, reversed operands (left extension), N padding in front of a strand, N bases."""
    from winnowmap_amd import synth
    rng = np.random.default_rng(11)
    ref = synth.make_reference(3, 60000, 21, repeat_frac=0.0)
    ref[1][1000:1040] = 4                                        # a run of N in the second contig
    fa = str(tmp_path / "ref.fa")
    synth.write_fasta(fa, ref)
    c = gpu.Context(0, 2 << 30)
    idx = gpu.Index(fa, None, k=15, w=50, n_threads=2)
    idx.upload(c)
    lens = [3000, 5000, 2500, 4000]
    reads = [rng.integers(0, 4, n).astype(np.uint8) for n in lens]
    reads[2][100:110] = 4
    codes = np.concatenate(reads)
    offs = np.concatenate([[0], np.cumsum(lens)])
    c.reads_upload(codes)
    sc = gpu.KswScore(2, -4, -1, 4, 2, 24, 1)
    n = 300
    pos = np.zeros(n, gpu.KSW_POS_DTYPE)
    pairs = []
    for i in range(n):
        ri = int(rng.integers(0, len(reads)))
        sub0 = int(rng.integers(0, lens[ri] // 2))                # a window of the read, like a stage-1 prefix
        L = int(rng.integers(600, lens[ri] - sub0 + 1))
        win = reads[ri][sub0:sub0 + L]
        both = np.concatenate([np.full(8, 4, np.uint8), win, np.where(win[::-1] < 4, 3 - win[::-1], 4).astype(np.uint8)])   # qseq0 with its padding
        ql, tl = int(rng.integers(1, 500)), int(rng.integers(1, 500))
        if i in (1, 150):                                         # degenerate jobs inside a batch (the reference returns from them at once, src/ksw2_extd2_sse.c:68):
            ql, tl = (0, tl) if i == 1 else (ql, 0)               # they must not disturb their neighbours' operands in the slab
        step = -1 if i % 3 == 0 else 1
        strand = int(rng.integers(0, 2))
        q_lo = int(rng.integers(0, L - ql + 1)) + strand * L
        if i % 17 == 0 and q_lo >= 4:                             # a few bases in front of a strand (mm_align1_inv, src/align.c:826-828)
            q_lo = strand * L - 3
        rid = int(rng.integers(0, 3))
        t_lo = int(rng.integers(900, 1100 - 0)) if (rid == 1 and i % 5 == 0) else int(rng.integers(0, 60000 - tl + 1))
        q = both[8 + q_lo:8 + q_lo + ql]
        t = ref[rid][t_lo:t_lo + tl]
        if step < 0:
            q, t = q[::-1], t[::-1]
        flag = [0, 0x08, 0x40, 0x40 | 0x02 | 0x80][i % 4]
        w = [751, 100, 30, 751][i % 4]
        pairs.append((q.copy(), t.copy(), dict(w=w, zdrop=400, end_bonus=-1 if i % 2 else 5, flag=flag)))
        has_n = bool((q >= 4).any() or (t >= 4).any())
        pos[i] = (offs[ri] + sub0, L, q_lo if step > 0 else q_lo + ql - 1, rid, t_lo if step > 0 else t_lo + tl - 1, ql, tl, w, 400, -1 if i % 2 else 5, flag, step, has_n, 0)
    jobs, seqs = gpu.pack_jobs(pairs)
    r0, p0 = c.ksw_batch(sc, jobs, seqs)
    r1, p1 = c.ksw_batch_pos(sc, pos)
    for k in r0.dtype.names:
        assert np.array_equal(r0[k], r1[k]), k
    assert np.array_equal(p0, p1)
    # the z-drop scan of mm_test_zdrop (src/align.c:32-66) on the device over the finished CIGARs (wm_ksw_batch_pos_zd, ksw_zdwalk_kernel) against the
    # host's compile of the same walk (csrc/cigar_walk.h through tests/host_harness — which tests/test_walks_vs_ref.py pins to the reference's own static
    # mm_test_zdrop, oracle/ref_align_shim.cpp) and, for the largest drop of a sample of jobs, against that reference function directly (binary search on
    # its verdict): flagged forward jobs carry their scan, everything else the neutral value; alignments and CIGARs are unchanged
    import ctypes as C
    from winnowmap_amd import build
    H = C.CDLL(build.build_harness())
    H.h_zdrop_walk.argtypes = [W.u8p, W.u8p, W.u32p, C.c_int] + [C.c_int] * 5 + [W.i32p]
    posz = pos.copy()
    fwd = (pos["step"] == 1) & ((pos["flag"] & 0x80) == 0)
    want = fwd & (np.arange(n) % 5 != 0)
    posz["flag"] = np.where(want, pos["flag"] | gpu.KSW_F_ZDWALK, pos["flag"])
    r2, p2, zd = c.ksw_batch_pos_zd(sc, posz)
    for k in r0.dtype.names:
        assert np.array_equal(r0[k], r2[k]), k
    assert np.array_equal(p0, p2)
    n_drop = n_ref = 0
    for i in range(n):
        if not want[i]:
            assert list(zd[i]) == [0, -1, -1, -1, -1], i
            continue
        q, t, _ = pairs[i]
        cig = np.ascontiguousarray(p2[r2["cig_off"][i]:r2["cig_off"][i] + r2["n_cigar"][i]])
        exp = np.zeros(5, np.int32)
        H.h_zdrop_walk(np.ascontiguousarray(q), np.ascontiguousarray(t), cig if len(cig) else np.zeros(1, np.uint32), len(cig), 2, -4, -1, 4, 2, exp)
        assert list(zd[i]) == list(exp), (i, list(zd[i]), list(exp))
        n_drop += int(exp[0] > 0)
        if W.have_ref() and n_ref < 40:      # the reference's verdict is "largest drop > zdrop" once its inversion test is off (MM_F_FOR_ONLY): bisect it
            R = W.ref()
            R.refshim_test_zdrop.argtypes = [C.c_int64] + [C.c_int] * 8 + [W.u8p, W.u8p, C.c_uint32, W.u32p, W.i8p]
            mat = W.simple_mat(2, 4, 1)
            lo, hi = -1, 1 << 22
            qq, tt, cc = np.ascontiguousarray(q), np.ascontiguousarray(t), cig if len(cig) else np.zeros(1, np.uint32)
            while hi - lo > 1:
                mid = (lo + hi) // 2
                if R.refshim_test_zdrop(0x100000, mid, mid, 4, 2, 5000, 40, 2, 80, qq, tt, len(cig), cc, mat):
                    lo = mid
                else:
                    hi = mid
            assert int(zd[i][0]) == hi, (i, int(zd[i][0]), hi)
            n_ref += 1
    assert int(want.sum()) >= 100 and n_drop >= 30 and (n_ref >= 30 or not W.have_ref()), (int(want.sum()), n_drop, n_ref)
    # positions outside the resident data are refused, not read
    bad = pos[:1].copy()
    bad["t_pos"] = 59990; bad["tlen"] = 100; bad["step"] = 1
    with pytest.raises(gpu.WmError):
        c.ksw_batch_pos(sc, bad)
    idx.close(); c.close()


# ---- the stripe-pipelined multi-wave kernels (csrc/ksw_stripe_kernel.h): every job of the suites above once more with the routing thresholds at
# one row, so that each runs on NWV wavefronts that hand rows to each other through LDS (geometry by traceback pitch), and natively wide hulls ----
def _default_routing():
    w16 = int(os.environ.get("WM_KSW_STRIPE16", 1)) & 3
    on = 0 if os.environ.get("WM_KSW_STRIPE") == "0" else 3 if w16 == 0 else 2 if w16 >= 2 else 1
    gpu.set_ksw_routing(on, int(os.environ.get("WM_KSW_STRIPE_ROWS4", 0)), int(os.environ.get("WM_KSW_STRIPE_ROWS8", 4096)))


# on = 3: the geometries of four / eight wavefronts (<2,4> <2,8> <4,8> <8,8>); on = 2: the sixteen-wavefront geometries (<1,16> <2,16>) wherever they fit;
# on = 1: the default routing (<2,16> for hulls of 1793..3840 lanes, else as 3)
@pytest.fixture(params=[3, 2], ids=["nwv4_8", "nwv16"])
def all_stripes(request):
    gpu.set_ksw_routing(request.param, 1, 1)
    yield
    _default_routing()


@pytest.fixture(params=[3, 2, 1], ids=["nwv4_8", "nwv16", "default"])
def stripe_geometries(request):
    gpu.set_ksw_routing(request.param, -1, -1)
    yield
    _default_routing()


@pytest.mark.parametrize("preset", [0, 1, 2, 3, 4])
def test_stripe_kernels_random_cases_all_flags(ctx, all_stripes, preset):
    cases = kswcases.make_cases(300 + preset, 240, max_len=900, preset=preset)
    bad = _run_group(ctx, cases)
    assert not bad, bad[:3]


def test_stripe_kernels_long_jobs_every_band(ctx, all_stripes):
    bad = _run_group(ctx, kswcases.stripe_cases(11, 300, 2600))
    assert not bad, bad[:3]


def test_stripe_kernels_wide_hulls(ctx, stripe_geometries):
    """hulls of 1 100 .. 7 000 lanes: geometries <2,8>, <4,8>, <8,8> by default routing — <1,16>, <2,16>, <8,8> with the sixteen-wavefront geometries
    (exact + z-drop, approximate maximum, unbanded, an N)"""
    from winnowmap_amd import synth
    rng = np.random.default_rng(21)
    cases = []
    for it, (L, w, flag, zd) in enumerate(((1300, 3001, 0x40, 400), (1700, 1501, 0x08, 400), (2500, -1, 0x08, 400), (3300, 3001, 0x40, 200), (3300, 3001, 0x00, 400),
                                          (5200, -1, 0x08, 400), (6800, -1, 0x40, 400), (2100, 1200, 0xC2, 100), (4000, 2500, 0x42, 400))):
        t = rng.integers(0, 4, L).astype(np.uint8)
        q = synth.mutate_codes(t, rng, 0.03, 0.03, 0.04)
        if it % 3 == 1:
            q = np.concatenate([q[:L // 2], rng.integers(0, 4, L // 3).astype(np.uint8)])      # runs off: z-drop
        if it == 4:
            t[L // 2] = 4
        cases.append(dict(q=q, t=t, a=2, b=4, q_=4, e=2, q2=24, e2=1, w=w, zdrop=zd, end_bonus=-1, flag=flag))
    bad = _run_group(ctx, cases)
    assert not bad, bad[:3]


# ---- the chained-workgroup kernels (csrc/ksw_chain_kernel.h): every wavefront of an alignment a workgroup of its own, row messages through a mailbox in
# HBM, tickets instead of block indices. mode 4 sends EVERY job there (two wavefronts for a tiny job), the default mode the wide and long ones ----
@pytest.fixture(params=[2, 4], ids=["bp2", "bp4"])
def all_chained(request):
    gpu.set_ksw_chain_routing(4, 1, request.param)
    yield
    _default_chain_routing()


@pytest.mark.parametrize("preset", [0, 1, 2, 3, 4])
def test_chain_kernels_random_cases_all_flags(ctx, all_chained, preset):
    cases = kswcases.make_cases(500 + preset, 240, max_len=900, preset=preset)
    bad = _run_group(ctx, cases)
    assert not bad, bad[:3]


def test_chain_kernels_long_jobs_every_band(ctx, all_chained):
    bad = _run_group(ctx, kswcases.stripe_cases(17, 300, 2600))                # (stripe_edge_cases draw a scoring preset per case: emulator tests only)
    assert not bad, bad[:3]


@pytest.mark.parametrize("bp", [2, 4])
def test_chain_kernels_wide_hulls_incl_beyond_the_stripe_kernels_reach(ctx, bp):
    """hulls of 1 100 .. 12 000 lanes with the default routing (mode 1): what the stripe classes served, and beyond 7 168 lanes what only ksw_pmulti_kernel<8,8>,
    ksw_block_kernel and ksw_generic_kernel could (exact + z-drop, approximate maximum, unbanded, an N)"""
    from winnowmap_amd import synth
    gpu.set_ksw_chain_routing(1, -1, bp)
    try:
        rng = np.random.default_rng(23)
        cases = []
        for it, (L, w, flag, zd) in enumerate(((1300, 3001, 0x40, 400), (1700, 1501, 0x08, 400), (2500, -1, 0x08, 400), (3300, 3001, 0x40, 200), (3300, 3001, 0x00, 400),
                                              (5200, -1, 0x08, 400), (6800, -1, 0x40, 400), (2100, 1200, 0xC2, 100), (4000, 2500, 0x42, 400),
                                              (7600, -1, 0x08, 400), (9100, -1, 0x40, 400), (12000, -1, 0x08, 400), (8200, 8000, 0x42, 300))):
            t = rng.integers(0, 4, L).astype(np.uint8)
            q = synth.mutate_codes(t, rng, 0.03, 0.03, 0.04)
            if it % 3 == 1:
                q = np.concatenate([q[:L // 2], rng.integers(0, 4, L // 3).astype(np.uint8)])      # runs off: z-drop
            if it == 4:
                t[L // 2] = 4
            cases.append(dict(q=q, t=t, a=2, b=4, q_=4, e=2, q2=24, e2=1, w=w, zdrop=zd, end_bonus=-1, flag=flag))
        bad = _run_group(ctx, cases)
        assert not bad, bad[:3]
    finally:
        _default_chain_routing()


def test_chain_kernels_a_launch_larger_than_the_chip(ctx):
    """900 wide jobs in one call = ~9 000 single-wavefront workgroups of one class, several times what the chip holds at once: tickets are taken in start
    order, consumers start after their producers have run into back-pressure, and every job must still come out right (checked against the oracle on a
    sample, against the one-wavefront-per-job result for all: results never depend on the routing)"""
    from winnowmap_amd import synth
    rng = np.random.default_rng(29)
    cases = []
    for it in range(900):
        L = int(rng.integers(1100, 2600))
        t = rng.integers(0, 4, L).astype(np.uint8)
        q = synth.mutate_codes(t, rng, 0.04, 0.03, 0.04)
        if it % 7 == 3:
            q = np.concatenate([q[:L // 3], rng.integers(0, 4, L // 2).astype(np.uint8)])
        cases.append(dict(q=q, t=t, a=2, b=4, q_=4, e=2, q2=24, e2=1, w=[3001, -1, 1501][it % 3], zdrop=[400, 200][it % 2], end_bonus=-1, flag=[0x40, 0x08, 0x00, 0x42][it % 4]))
    sc = gpu.KswScore(2, -4, -1, 4, 2, 24, 1)
    jobs, seqs = gpu.pack_jobs([(c["q"], c["t"], dict(w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])) for c in cases])
    gpu.set_ksw_chain_routing(1, -1, 2)
    try:
        res, pool = ctx.ksw_batch(sc, jobs, seqs)
    finally:
        gpu.set_ksw_chain_routing(0, -1, -1)
    res0, pool0 = ctx.ksw_batch(sc, jobs, seqs)              # the stripe / register kernels
    for i in range(len(cases)):
        assert all(int(res[i][k]) == int(res0[i][k]) for k in W.EZ_FIELDS), (i, res[i], res0[i])
        assert np.array_equal(pool[res[i]["cig_off"]:res[i]["cig_off"] + res[i]["n_cigar"]], pool0[res0[i]["cig_off"]:res0[i]["cig_off"] + res0[i]["n_cigar"]]), i
    for i in range(0, len(cases), 60):
        c = cases[i]
        o = W.o_ksw_extd2(c["q"], c["t"], mat=W.simple_mat(2, 4, 1), q=4, e=2, q2=24, e2=1, w=c["w"], zdrop=c["zdrop"], end_bonus=-1, flag=c["flag"])
        assert all(int(res[i][k]) == o[k] for k in W.EZ_FIELDS), (i, res[i], o)


def test_two_alignments_per_wavefront_equal_one_per_wavefront_and_the_oracle(ctx):
    """ksw_dual_kernel (round 6; off by default, wm_ksw_set_dual / WM_KSW_DUAL=1): two gap fills per wavefront. Batches of gap fills of every size (so that the
    size-sorted class lists pair neighbours, an odd list leaves one job alone, short jobs finish long before their partners) against the oracle, and the same
    batch with one alignment per wavefront must give identical results."""
    was = gpu.ksw_dual_enabled()
    try:
        for seed in (3, 4, 5):
            sc, cases = None, []
            for nc, s, jobs in kswcases.dual_pairs(seed, 400, ncs=(8, 16)):
                sc = sc or s
                cases += [dict(j, **sc) for j in jobs]           # one scoring set per batch
            cases = cases[:701]                                   # (odd: the last job of a class list is alone in its wavefront)
            gpu.set_ksw_dual(1)
            bad = _run_group(ctx, cases)
            assert not bad, bad[:3]
            gpu.set_ksw_dual(0)
            bad = _run_group(ctx, cases)
            assert not bad, bad[:3]
    finally:
        gpu.set_ksw_dual(was)
