"""GPU parity of wm_window_batch — sketch → seed → sort → chain → extraction in one device call — against the oracle's functions applied
one after the other on the host: o_sketch (mm_sketch), collect_seed_hits restated over the reference's / product's index, o_radix_sort_128x
(the reference's unstable sort incl. its tie permutation) and o_chain_dp (mm_chain_dp). Sequences come resident (wm_reads_upload), staged,
or not at all (handed-in anchors only); handed-in anchors precede the seeded ones as in stage 2 (src/map.c:818-833)."""
import ctypes as C
import numpy as np
import pytest
import wmtest as W
from winnowmap_amd import gpu, synth
from test_aux_gpu import env, _windows, M128  # noqa: F401

pytestmark = pytest.mark.gpu

JOB = np.dtype([("seq_off", np.int64), ("stage_off", np.uint64), ("pre_off", np.uint64), ("len", np.int32), ("n_pre", np.int32),
                ("par", np.int32, 8), ("gs", np.float32), ("is_cdna", np.int32)], align=False)
RES = np.dtype([("n_anchors", np.int32), ("rep_len", np.int32), ("n_mini", np.int32), ("n_u", np.int32), ("n_v", np.int32), ("u_off", np.uint32), ("a_off", np.uint32)])


def _index_get(L, idx, key):
    t = C.c_int()
    if L._wm_ref_index is not None:
        R, mi = L._wm_ref_index
        pbuf = np.zeros(8192, np.uint64)
        t.value = R.refshim_idx_get(mi, key, pbuf, len(pbuf))
        return pbuf, t.value
    p = L.wm_index_get(idx._h, key, C.byref(t))
    return p, t.value


def expected_seeds(L, idx, s, mx, my, max_occ=5000):
    ex, ey, rep_st, rep_en, rep = [], [], 0, 0, 0
    for j in range(len(mx)):
        x, y = int(mx[j]), int(my[j])
        p, t = _index_get(L, idx, x >> 8)
        qpos, span = y & 0xffffffff, x & 0xff
        if t >= max_occ:
            en = (qpos >> 1) + 1
            st = en - span
            if st > rep_en:
                rep += rep_en - rep_st
                rep_st, rep_en = st, en
            else:
                rep_en = en
            continue
        tand = (j > 0 and int(mx[j - 1]) >> 8 == x >> 8) or (j < len(mx) - 1 and int(mx[j + 1]) >> 8 == x >> 8)
        for q in range(t):
            r = int(p[q])
            rpos = (r & 0xffffffff) >> 1
            if (r & 1) == (qpos & 1):
                X, Y = (r & 0xffffffff00000000) | rpos, span << 32 | qpos >> 1
            else:
                X, Y = 1 << 63 | (r & 0xffffffff00000000) | rpos, span << 32 | (len(s) - ((qpos >> 1) + 1 - span) - 1)
            if tand:
                Y |= 1 << 42
            ex.append(X)
            ey.append(Y)
    rep += rep_en - rep_st
    return np.array(ex, np.uint64), np.array(ey, np.uint64), rep


def test_window_batch_equals_the_oracle_stage_by_stage(env):
    ctx, idx, ref, bloom, L = env
    L.wm_window_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int64, C.c_void_p,
                                  C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    assert JOB.itemsize == 72 and RES.itemsize == 28
    rng = np.random.default_rng(9)
    seqs = _windows(ref)
    n_w = len(seqs)
    resident = np.concatenate(seqs)
    ctx.reads_upload(resident)
    roff = np.concatenate([[0], np.cumsum([len(s) for s in seqs])[:-1]])
    # job list: every window three times — resident, staged, staged with handed-in anchors — plus anchor-only jobs and an empty one
    jobs, pres, stage = [], [], []
    spos = ppos = 0
    PRM = ((5000, 1000, 5000, 500, 25, 5000, 3, 40), (16000, 1000, 16000, 2000, 25, 5000, 3, 40))

    def some_anchors(k):
        x = rng.integers(0, 2, k).astype(np.uint64) << np.uint64(63) | rng.integers(0, 2, k).astype(np.uint64) << np.uint64(32) | rng.integers(0, 300000, k).astype(np.uint64)
        if k > 4:
            x[rng.integers(0, k, k // 4)] = x[rng.integers(0, k, k // 4)]          # ties
        x = np.sort(x)
        y = rng.integers(0, 14000, k).astype(np.uint64) | np.uint64(15 << 32)
        return x, y

    for i, s in enumerate(seqs):
        prm = PRM[i % 2]
        jobs.append(dict(kind="resident", w=i, seq_off=int(roff[i]), stage_off=0, len=len(s), pre=None, prm=prm))
        jobs.append(dict(kind="staged", w=i, seq_off=-1, stage_off=spos, len=len(s), pre=None, prm=prm))
        stage.append(s); spos += len(s)
        if i % 3 == 0:
            px, py = some_anchors(int(rng.integers(1, 200)))
            jobs.append(dict(kind="pre+seq", w=i, seq_off=int(roff[i]), stage_off=0, len=len(s), pre=(px, py), prm=prm))
    for k in (1, 70, 900):
        px, py = some_anchors(k)
        jobs.append(dict(kind="pre", w=-1, seq_off=-2, stage_off=0, len=0, pre=(px, py), prm=PRM[1]))
    jobs.append(dict(kind="empty", w=-1, seq_off=-2, stage_off=0, len=0, pre=None, prm=PRM[0]))
    n = len(jobs)
    J = np.zeros(n, JOB)
    for i, j in enumerate(jobs):
        J[i]["seq_off"] = j["seq_off"]; J[i]["stage_off"] = j["stage_off"]; J[i]["len"] = j["len"]
        J[i]["par"] = j["prm"]; J[i]["gs"] = 1.0
        if j["pre"] is not None:
            J[i]["pre_off"] = ppos; J[i]["n_pre"] = len(j["pre"][0]); ppos += len(j["pre"][0])
            a = np.zeros(len(j["pre"][0]), M128); a["x"], a["y"] = j["pre"]
            pres.append(a)
    stage_all = np.concatenate(stage)
    pre_all = np.concatenate(pres)
    res = np.zeros(n, RES)
    ucap, acap = 1 << 20, 1 << 22
    up = np.zeros(ucap, np.uint64); ap = np.zeros(acap, M128)
    uu, au = C.c_size_t(), C.c_size_t()
    # too small pools are reported with the sizes needed
    rc = L.wm_window_batch(ctx._h, n, J.ctypes.data, stage_all.ctypes.data, stage_all.nbytes, pre_all.ctypes.data, len(pre_all), 5000, 0, res.ctypes.data,
                           up.ctypes.data, 1, C.byref(uu), ap.ctypes.data, 1, C.byref(au))
    assert rc == -3 and uu.value > 1 and au.value > 1
    rc = L.wm_window_batch(ctx._h, n, J.ctypes.data, stage_all.ctypes.data, stage_all.nbytes, pre_all.ctypes.data, len(pre_all), 5000, 0, res.ctypes.data,
                           up.ctypes.data, ucap, C.byref(uu), ap.ctypes.data, acap, C.byref(au))
    assert rc == 0, L.wm_last_error()
    cache = {}
    n_ties = n_chained = 0
    for i, j in enumerate(jobs):
        if j["w"] >= 0:
            if j["w"] not in cache:
                s = seqs[j["w"]]
                mx, my = W.o_sketch(bytes(s), 50, 15, rid=0, bloom=bloom)
                ex, ey, rep = expected_seeds(L, idx, s, mx, my)
                sx, sy = W.o_radix_sort_128x(ex, ey)
                cache[j["w"]] = (len(mx), sx, sy, rep)
            n_mini, sx, sy, rep = cache[j["w"]]
        else:
            n_mini, sx, sy, rep = 0, np.zeros(0, np.uint64), np.zeros(0, np.uint64), 0
        if j["pre"] is not None:
            ax, ay = np.concatenate([j["pre"][0], sx]), np.concatenate([j["pre"][1], sy])
            if j["len"] > 0:
                ax, ay = W.o_radix_sort_128x(ax, ay)
        else:
            ax, ay = sx, sy
        r = res[i]
        assert r["n_anchors"] == len(ax) and r["rep_len"] == rep and r["n_mini"] == n_mini, (i, j["kind"], r, len(ax), rep, n_mini)
        p = j["prm"]
        if len(ax):
            ou, obx, oby = W.o_chain_dp(ax, ay, max_dist_x=p[0], min_dist_x=p[1], max_dist_y=p[2], bw=p[3])
        else:
            ou = obx = oby = np.zeros(0, np.uint64)
        gu = up[int(r["u_off"]):int(r["u_off"]) + int(r["n_u"])]
        ga = ap[int(r["a_off"]):int(r["a_off"]) + int(r["n_v"])]
        assert np.array_equal(gu, ou) and np.array_equal(ga["x"], obx) and np.array_equal(ga["y"], oby), (i, j["kind"], j["w"], len(ax), r)
        n_ties += int(len(ax) > 64 and len(np.unique(ax)) < len(ax))
        n_chained += int(len(ou) > 0)
    assert n_ties >= 5 and n_chained >= 50, (n_ties, n_chained)       # (jobs beyond the insertion-sort size whose keys tie: the permutation matters)
    assert uu.value == int(res["n_u"].sum()) and au.value == int(res["n_v"].sum())


def test_window_batch_large_jobs_global_memory_paths(env):
    """anchor sets beyond the LDS classes (> 4096 anchors: sort and extraction in global memory; dense tandem arrays: the 8-wave fill)"""
    ctx, idx, ref, bloom, L = env
    L.wm_window_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int64, C.c_void_p,
                                  C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    rng = np.random.default_rng(17)
    sets = []
    nn = 6000
    xs = np.sort(rng.integers(0, 250000, nn)).astype(np.uint64)
    ys = (xs.astype(np.int64) + rng.integers(-30, 30, nn)).clip(0, None).astype(np.uint64)
    noise = rng.random(nn) < 0.2
    ys[noise] = rng.integers(0, 250000, int(noise.sum())).astype(np.uint64)
    sets.append((xs, ys | np.uint64(15 << 32)))
    qpos = np.arange(40, 4000, 57)
    rpos = np.arange(1000, 9000, 171)
    X, Y = np.meshgrid(rpos, qpos)
    xs = X.ravel().astype(np.uint64) + rng.integers(0, 3, X.size).astype(np.uint64)
    ys = Y.ravel().astype(np.uint64)
    o = np.argsort(xs, kind="stable")
    sets.append((xs[o], ys[o] | np.uint64(15 << 32)))
    sets.append((np.sort(rng.integers(0, 3000, 1500)).astype(np.uint64), rng.integers(0, 3000, 1500).astype(np.uint64) | np.uint64(15 << 32)))      # many ties, LDS class
    parts = []
    J = np.zeros(len(sets), JOB)
    pos = 0
    for i, (x, y) in enumerate(sets):
        sx, sy = W.o_radix_sort_128x(x, y)
        a = np.zeros(len(sx), M128); a["x"], a["y"] = sx, sy
        parts.append(a)
        J[i]["seq_off"] = -2; J[i]["pre_off"] = pos; J[i]["n_pre"] = len(a); pos += len(a)
        J[i]["par"] = (5000, 1000, 5000, 500, 25, 5000, 3, 40); J[i]["gs"] = 1.0
    pre_all = np.concatenate(parts)
    res = np.zeros(len(sets), RES)
    up = np.zeros(1 << 18, np.uint64); ap = np.zeros(1 << 18, M128)
    uu, au = C.c_size_t(), C.c_size_t()
    dummy = np.zeros(8, np.uint8)
    rc = L.wm_window_batch(ctx._h, len(sets), J.ctypes.data, dummy.ctypes.data, 0, pre_all.ctypes.data, len(pre_all), 5000, 0, res.ctypes.data,
                           up.ctypes.data, len(up), C.byref(uu), ap.ctypes.data, len(ap), C.byref(au))
    assert rc == 0, L.wm_last_error()
    for i, a in enumerate(parts):
        ou, obx, oby = W.o_chain_dp(a["x"], a["y"])
        r = res[i]
        gu = up[int(r["u_off"]):int(r["u_off"]) + int(r["n_u"])]
        ga = ap[int(r["a_off"]):int(r["a_off"]) + int(r["n_v"])]
        assert np.array_equal(gu, ou) and np.array_equal(ga["x"], obx) and np.array_equal(ga["y"], oby), (i, len(a), r)


def test_window_batch_giant_seeded_jobs_workgroup_sort_and_tie_fallback():
    """reads inside a high-copy repeat: every minimizer hits hundreds of reference positions -> anchor sets of 10^4 .. 10^5 per window, sorted by
    the workgroup kernel (win_bigsort_block); a read that holds the element twice makes every key tie -> the literal replay of the reference's
    permutation (win_sort_wave) on a giant job. Expectation as above: the oracle's functions stage by stage."""
    import tempfile
    rng = np.random.default_rng(31)
    elem = rng.integers(0, 4, 1500).astype(np.uint8)
    contig = rng.integers(0, 4, 1500000).astype(np.uint8)
    pos = np.sort(rng.choice(np.arange(2000, 1490000, 4000), 260, replace=False))
    for p in pos:
        contig[p:p + len(elem)] = elem
    tmp = tempfile.mkdtemp()
    synth.write_fasta(tmp + "/ref.fa", [contig])
    ctx = gpu.Context(0, 6 << 30)
    idx = gpu.Index(tmp + "/ref.fa", None, k=15, w=50)
    idx.upload(ctx)
    L = gpu.lib()
    L.wm_index_get.restype = C.POINTER(C.c_uint64)
    L.wm_index_get.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_int)]
    L._wm_ref_index = None
    L.wm_window_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int64, C.c_void_p,
                                  C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    flank = lambda n: rng.integers(0, 4, n).astype(np.uint8)
    seqs = [np.concatenate([flank(300), elem, flank(200)]),                                  # one copy: ~ 60 minimizers x 260 hits, distinct keys
            np.concatenate([flank(100), elem, flank(50), elem, flank(100)]),                 # the element twice: every reference position is hit twice
            np.concatenate([flank(100), elem[:700], flank(30), synth.revcomp_codes(elem), flank(10)]),   # both strands
            contig[pos[3] - 400:pos[3] + 1900].copy()]
    J = np.zeros(len(seqs), JOB)
    spos = 0
    for i, s in enumerate(seqs):
        J[i]["seq_off"] = -1; J[i]["stage_off"] = spos; J[i]["len"] = len(s); spos += len(s)
        J[i]["par"] = (5000, 1000, 5000, 500, 25, 5000, 3, 40); J[i]["gs"] = 1.0
    stage = np.concatenate(seqs)
    res = np.zeros(len(seqs), RES)
    up = np.zeros(1 << 18, np.uint64); ap = np.zeros(1 << 20, M128)
    uu, au = C.c_size_t(), C.c_size_t()
    dummy = np.zeros(1, M128)
    rc = L.wm_window_batch(ctx._h, len(seqs), J.ctypes.data, stage.ctypes.data, stage.nbytes, dummy.ctypes.data, 0, 5000, 0, res.ctypes.data,
                           up.ctypes.data, len(up), C.byref(uu), ap.ctypes.data, len(ap), C.byref(au))
    assert rc == 0, L.wm_last_error()
    n_giant = n_tied = 0
    for i, s in enumerate(seqs):
        mx, my = W.o_sketch(bytes(s), 50, 15, rid=0, bloom=None)
        ex, ey, rep = expected_seeds(L, idx, s, mx, my)
        sx, sy = W.o_radix_sort_128x(ex, ey)
        r = res[i]
        assert r["n_anchors"] == len(sx) and r["rep_len"] == rep, (i, r, len(sx), rep)
        ou, obx, oby = W.o_chain_dp(sx, sy)
        gu = up[int(r["u_off"]):int(r["u_off"]) + int(r["n_u"])]
        ga = ap[int(r["a_off"]):int(r["a_off"]) + int(r["n_v"])]
        assert np.array_equal(gu, ou) and np.array_equal(ga["x"], obx) and np.array_equal(ga["y"], oby), (i, len(sx), r)
        n_giant += int(len(sx) > 4096)
        n_tied += int(len(sx) > 4096 and len(np.unique(sx)) < len(sx))
    assert n_giant >= 3 and n_tied >= 1, (n_giant, n_tied)
    idx.close(); ctx.close()
