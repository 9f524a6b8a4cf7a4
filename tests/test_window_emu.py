"""The fused window kernels (winnowmap_amd/csrc/window_kernel.h) on the host wavefront emulator, against the oracle:
win_seed_wave = collect_seed_hits before its sort, win_sort_wave = radix_sort_128x WITH the reference's tie permutation,
win_plan_wave = avg_qspan, win_extract_wave = mm_chain_dp after the fill (src/chain.c:89-165)."""
import ctypes as C
import numpy as np
import pytest
import wmtest as W
from winnowmap_amd import build
from test_kernels_emu import small_index, _expected_anchors, _load_emu  # noqa: F401  (fixtures / helpers)


@pytest.fixture(scope="module")
def emu():
    E = _load_emu()
    E.emu_win_seed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, W.u64p, W.u64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, W.u64p, W.u64p, W.u64p, W.u64p, C.c_int, W.i32p]
    E.emu_win_sort.argtypes = [C.c_int, W.u64p, W.u64p, C.c_int]
    E.emu_win_plan.argtypes = [C.c_int, W.u64p, W.u64p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    E.emu_win_extract.argtypes = [C.c_int, W.u64p, W.u64p, W.i32p, W.i32p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), W.u64p]
    E.emu_win_bigsort.argtypes = [C.c_int, W.u64p, W.u64p, C.c_int]
    E.emu_win_small.argtypes = [C.c_int, C.c_int, C.c_int, W.u64p, W.u64p] + [C.c_int] * 8 + [C.c_float, C.POINTER(C.c_int), W.u64p]
    return E


def _anchor_like_keys(rng, n, tie_frac, n_rid=3, span=200000, both_strands=True):
    """keys shaped like anchors (strand | contig | position) with a chosen fraction of duplicated keys"""
    rid = rng.integers(0, n_rid, n).astype(np.uint64)
    pos = rng.integers(0, span, n).astype(np.uint64)
    rev = (rng.integers(0, 2, n).astype(np.uint64) if both_strands else np.zeros(n, np.uint64))
    x = rev << np.uint64(63) | rid << np.uint64(32) | pos
    n_dup = int(n * tie_frac)
    if n_dup:
        src = rng.integers(0, n, n_dup)
        dst = rng.integers(0, n, n_dup)
        x[dst] = x[src]
    return x


@pytest.mark.parametrize("glob", [0, 1])
def test_sort_reproduces_the_reference_tie_permutation(emu, glob):
    rng = np.random.default_rng(11 + glob)
    sizes = [0, 1, 2, 3, 17, 63, 64, 65, 66, 90, 127, 128, 129, 200, 257, 700, 1500, 4100]
    n_cases = 0
    for n in sizes:
        for tie in (0.0, 0.1, 0.6):
            for shape in range(3):
                if shape == 0:
                    x = _anchor_like_keys(rng, n, tie)
                elif shape == 1:      # one contig, one strand, a narrow window: deep recursion into the low position bytes
                    x = _anchor_like_keys(rng, n, tie, n_rid=1, span=3000, both_strands=False)
                else:                 # every byte varies (chain scores / generic 64-bit keys)
                    x = rng.integers(0, 1 << 63, n, dtype=np.int64).astype(np.uint64)
                    if n and tie:
                        x[rng.integers(0, n, int(n * tie))] = x[0]
                y = np.arange(n, dtype=np.uint64)            # payload = original index: the permutation itself is compared
                ex, ey = W.o_radix_sort_128x(x, y)
                gx, gy = x.copy(), y.copy()
                emu.emu_win_sort(n, gx, gy, glob)
                assert np.array_equal(gx, ex) and np.array_equal(gy, ey), (n, tie, shape, glob)
                n_cases += 1
    # already sorted input (stage 2 hands in sorted anchors) and all-equal keys
    for n in (65, 300, 1000):
        x = np.sort(_anchor_like_keys(rng, n, 0.3)); y = np.arange(n, dtype=np.uint64)
        ex, ey = W.o_radix_sort_128x(x, y)
        gx, gy = x.copy(), y.copy()
        emu.emu_win_sort(n, gx, gy, glob)
        assert np.array_equal(gx, ex) and np.array_equal(gy, ey)
        x = np.full(n, 12345 << 20, np.uint64)
        ex, ey = W.o_radix_sort_128x(x, y)
        gx, gy = x.copy(), y.copy()
        emu.emu_win_sort(n, gx, gy, glob)
        assert np.array_equal(gy, ey)
    assert n_cases >= 150


def test_seed_plan_sort_extract_chain_on_real_windows(emu, small_index):
    """windows of reads through the whole device-side sequence (the fill itself is covered by test_kernels_emu): anchors incl. handed-in
    ones, rep_len, the sort, avg_qspan, and chains + regrouped anchors equal to the oracle's mm_chain_dp"""
    S = small_index
    H = S["H"]
    windows = [r[st:st + 2000].copy() for r in S["reads"][:2] for st in range(0, 14000, 3500)] + [S["reads"][2], S["reads"][3]]
    n_chain = 0
    n_small = [0]
    for wi, s in enumerate(windows):
        mx, my = W.o_sketch(bytes(s), 50, 15, bloom=S["bloom"])
        ex, ey, rep = _expected_anchors(S, mx, my, len(s))
        # handed-in anchors (stage 2, src/map.c:818-826): a sorted set that precedes the seeded ones
        n_pre = [0, 5, 70][wi % 3]
        px = np.sort(_anchor_like_keys(np.random.default_rng(wi), n_pre, 0.2)); py = np.arange(n_pre, dtype=np.uint64) + np.uint64(15 << 32)
        cap = 200000
        ax = np.zeros(cap, np.uint64); ay = np.zeros(cap, np.uint64); res = np.zeros(3, np.int32)
        emu.emu_win_seed(C.cast(S["hk"], C.c_void_p), C.cast(S["hv"], C.c_void_p), C.cast(S["P"], C.c_void_p), S["hbits"], mx, my, len(mx), len(s), 5000, 0,
                         n_pre, px, py, ax, ay, cap, res)
        n = int(res[0])
        assert res[2] == 0 and n == n_pre + len(ex) and res[1] == rep
        assert np.array_equal(ax[:n_pre], px) and np.array_equal(ay[:n_pre], py)
        assert np.array_equal(ax[n_pre:n], ex) and np.array_equal(ay[n_pre:n], ey)
        # the pool is too small: flagged, nothing written
        res2 = np.zeros(3, np.int32)
        emu.emu_win_seed(C.cast(S["hk"], C.c_void_p), C.cast(S["hv"], C.c_void_p), C.cast(S["P"], C.c_void_p), S["hbits"], mx, my, len(mx), len(s), 5000, 0,
                         n_pre, px, py, ax.copy(), ay.copy(), max(0, n - 1), res2)
        assert (res2[2] == 2 and res2[0] == 0) or n == 0
        if len(ex) == 0:
            continue
        sx, sy = W.o_radix_sort_128x(ex, ey)
        gx, gy = ex.copy(), ey.copy()
        emu.emu_win_sort(len(ex), gx, gy, wi & 1)
        assert np.array_equal(gx, sx) and np.array_equal(gy, sy)
        avg = C.c_float(); kl = C.c_int()
        emu.emu_win_plan(len(sx), sx, sy, 5000, C.byref(avg), C.byref(kl))
        assert avg.value == H.h_avg_qspan(len(sx), sy) and kl.value == (3 if len(sx) <= 256 else 2 if len(sx) <= 1024 else kl.value)
        prm = [dict(max_dist_x=5000, min_dist_x=1000, max_dist_y=5000, bw=500), dict(max_dist_x=16000, min_dist_x=1000, max_dist_y=16000, bw=2000)][wi % 2]
        # the same job through the one-wavefront LDS path (at most 256 anchors): unsorted seeds (+ handed-in anchors in front) -> chains
        for npre_s in (0, 9):
            jx, jy = ex.copy(), ey.copy()
            if npre_s:
                jx, jy = np.concatenate([px[:npre_s], jx]), np.concatenate([py[:npre_s], jy])
            if 0 < len(jx) <= 256 and (not npre_s or n_pre >= npre_s):
                if npre_s:
                    wx, wy = W.o_radix_sort_128x(np.concatenate([px[:npre_s], sx]), np.concatenate([py[:npre_s], sy]))
                else:
                    wx, wy = sx, sy
                ou, obx, oby = W.o_chain_dp(wx, wy, **prm)
                nu = C.c_int(); u = np.zeros(len(jx), np.uint64)
                nv = emu.emu_win_small(len(jx), npre_s, 1, jx, jy, prm["max_dist_x"], prm["min_dist_x"], prm["max_dist_y"], prm["bw"], 25, 5000, 3, 40, 1.0, C.byref(nu), u)
                assert nu.value == len(ou) and nv == len(obx) and np.array_equal(u[:nu.value], ou) and np.array_equal(jx[:nv], obx) and np.array_equal(jy[:nv], oby), (wi, npre_s)
                n_small[0] += 1
        nn = len(sx)
        fa = np.zeros(nn, np.int32); pa = np.zeros(nn, np.int32); va = np.zeros(nn, np.int32)
        emu.emu_chain_fill(nn, sx, sy, prm["max_dist_x"], prm["min_dist_x"], prm["max_dist_y"], prm["bw"], 25, 5000, avg.value, 1.0, fa, pa, va)
        for min_cnt, min_sc in ((3, 40), (1, 10), (5, 100)):
            ou, obx, oby = W.o_chain_dp(sx, sy, min_cnt=min_cnt, min_sc=min_sc, **prm)
            for glob in (0, 1):
                bx, by = sx.copy(), sy.copy()
                nu = C.c_int(); u = np.zeros(nn, np.uint64)
                nv = emu.emu_win_extract(nn, bx, by, fa, pa, min_cnt, min_sc, glob, C.byref(nu), u)
                assert nu.value == len(ou) and nv == len(obx), (wi, min_cnt, min_sc, glob, nu.value, len(ou), nv, len(obx))
                assert np.array_equal(u[:nu.value], ou) and np.array_equal(bx[:nv], obx) and np.array_equal(by[:nv], oby), (wi, min_cnt, min_sc, glob)
                n_chain += 1
    assert n_chain >= 40 and n_small[0] >= 8, (n_chain, n_small)


def test_extract_on_synthetic_forests(emu):
    """f / p arrays that are not the output of a fill but exercise the extraction's corners: many chain ends sharing prefixes (the shared-anchor
    cut), more than 64 chains (the w sort leaves the rank-sort path, ties in the first anchors' x), scores below the thresholds"""
    rng = np.random.default_rng(5)
    for it in range(30):
        n = int(rng.integers(1, 900))
        x = np.sort(_anchor_like_keys(rng, n, 0.3, n_rid=2, span=5000)); y = rng.integers(0, 1 << 20, n).astype(np.uint64) | np.uint64(15 << 32)
        p = np.full(n, -1, np.int32); f = np.zeros(n, np.int32)
        for i in range(n):
            if i and rng.random() < 0.8:
                p[i] = int(rng.integers(max(0, i - 40), i))
                f[i] = f[p[i]] + int(rng.integers(-6, 20))
            else:
                f[i] = int(rng.integers(5, 30))
        f = np.maximum(f, 1).astype(np.int32)
        # the oracle's extraction on the same f / p: restated here in numpy-free python straight from src/chain.c:89-165
        eu, ebx, eby = _py_extract(x, y, f, p, 3, 25)
        for glob in (0, 1):
            bx, by = x.copy(), y.copy()
            nu = C.c_int(); u = np.zeros(n, np.uint64)
            nv = emu.emu_win_extract(n, bx, by, f, p, 3, 25, glob, C.byref(nu), u)
            assert nu.value == len(eu) and nv == len(ebx), (it, n, glob)
            assert np.array_equal(u[:nu.value], eu) and np.array_equal(bx[:nv], ebx) and np.array_equal(by[:nv], eby), (it, n, glob)


def _py_extract(x, y, f, p, min_cnt, min_sc):
    n = len(x)
    v = [0] * n
    for i in range(n):
        v[i] = v[p[i]] if p[i] >= 0 and v[p[i]] > f[i] else int(f[i])
    t = [0] * n
    for i in range(n):
        if p[i] >= 0:
            t[p[i]] = 1
    z = []
    for i in range(n):
        if t[i] == 0 and v[i] >= min_sc:
            j = i
            while j >= 0 and f[j] < v[j]:
                j = int(p[j])
            if j < 0:
                j = i
            z.append(int(f[j]) << 32 | j)
    if not z:
        return np.zeros(0, np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.uint64)
    z.sort()
    z.reverse()
    t = [0] * n
    order, u = [], []
    for zi in z:
        n0 = len(order)
        j = zi & 0xffffffff
        while True:
            order.append(j); t[j] = 1; j = int(p[j])
            if not (j >= 0 and t[j] == 0):
                break
        cnt = len(order) - n0
        ok = False
        if j < 0:
            if cnt >= min_cnt:
                u.append((zi >> 32) << 32 | cnt); ok = True
        elif (zi >> 32) - int(f[j]) >= min_sc:
            if cnt >= min_cnt:
                u.append(((zi >> 32) - int(f[j])) << 32 | cnt); ok = True
        if not ok:
            del order[n0:]
    bx, by, wx, wy = [], [], [], []
    k = 0
    for i, ui in enumerate(u):
        ni = ui & 0xffffffff
        for j in range(ni):
            src = order[k + ni - 1 - j]
            bx.append(int(x[src])); by.append(int(y[src]))
        wx.append(bx[k]); wy.append(k << 32 | i)
        k += ni
    sx, sy = W.o_radix_sort_128x(np.array(wx, np.uint64), np.array(wy, np.uint64))
    ou, ox, oy = [], [], []
    for i in range(len(u)):
        j = int(sy[i]) & 0xffffffff; st = int(sy[i]) >> 32; cnt = u[j] & 0xffffffff
        ou.append(u[j]); ox += bx[st:st + cnt]; oy += by[st:st + cnt]
    return np.array(ou, np.uint64), np.array(ox, np.uint64), np.array(oy, np.uint64)


@pytest.mark.parametrize("nwv", [1, 3, 8])
def test_workgroup_sort_of_large_anchor_sets_is_the_stable_sort_and_flags_ties(emu, nwv):
    """win_bigsort_block: a stable LSD radix sort by a whole workgroup. Without ties it is the reference's order (any exact sort is); ties are
    reported so that the caller replays the reference's unstable permutation instead."""
    rng = np.random.default_rng(50 + nwv)
    for n in (0, 1, 2, 63, 64, 65, 500, 4097, 9000, 20011):
        for tie in (0.0, 0.05):
            for shape in range(2):
                x = _anchor_like_keys(rng, n, tie) if shape == 0 else rng.integers(0, 1 << 62, n, dtype=np.int64).astype(np.uint64)
                if shape == 1 and tie and n > 3:
                    x[n // 2] = x[n // 3]
                if tie == 0.0 and n:
                    x = np.unique(x); rng.shuffle(x)          # distinct keys
                m = len(x)
                y = np.arange(m, dtype=np.uint64)
                o = np.argsort(x, kind="stable")
                gx, gy = x.copy(), y.copy()
                flag = emu.emu_win_bigsort(m, gx, gy, nwv)
                assert flag in (0, 1), flag
                assert np.array_equal(gx, x[o]) and np.array_equal(gy, y[o]), (n, tie, shape, nwv)
                assert flag == int(m > 1 and len(np.unique(x)) < m)
                if flag == 0:
                    ex, ey = W.o_radix_sort_128x(x, y)
                    assert np.array_equal(gx, ex) and np.array_equal(gy, ey)


def test_chain_fill_with_the_cdna_gap_cost_matches_the_oracle(emu, small_index):
    """splice mode (is_cdna, src/chain.c:69-74): reference gaps cost min(linear, log). Transcript-like anchor sets: exons of a read placed
    with intron-sized jumps on the reference."""
    rng = np.random.default_rng(77)
    n_cases = 0
    for it in range(12):
        xs, ys = [], []
        rpos, qpos = 1000, 20
        for ex in range(int(rng.integers(2, 7))):
            for k in range(int(rng.integers(3, 40))):
                step = int(rng.integers(5, 40))
                rpos += step; qpos += step + int(rng.integers(-2, 3))
                xs.append(rpos); ys.append(qpos)
            rpos += int(rng.integers(80, 60000))          # an intron
        xs = np.array(xs, np.uint64); ys = np.array(ys, np.uint64) | np.uint64(15 << 32)
        noise = rng.integers(0, 200000, 10).astype(np.uint64)
        xs = np.concatenate([xs, noise]); ys = np.concatenate([ys, rng.integers(0, 3000, 10).astype(np.uint64) | np.uint64(15 << 32)])
        sx, sy = W.o_radix_sort_128x(xs, ys)
        n = len(sx)
        prm = dict(max_dist_x=200000, min_dist_x=1000, max_dist_y=2000, bw=200000)
        avg = C.c_float(); kl = C.c_int()
        emu.emu_win_plan(n, sx, sy, prm["max_dist_x"], C.byref(avg), C.byref(kl))
        fa = np.zeros(n, np.int32); pa = np.zeros(n, np.int32); va = np.zeros(n, np.int32)
        emu.emu_chain_fill(n, sx, sy, prm["max_dist_x"], prm["min_dist_x"], prm["max_dist_y"], prm["bw"], 25 | 0x8000, 5000, avg.value, 1.0, fa, pa, va)
        ou, obx, oby = W.o_chain_dp(sx, sy, is_cdna=1, **prm)
        ou0, _, _ = W.o_chain_dp(sx, sy, is_cdna=0, **prm)
        bx, by = sx.copy(), sy.copy()
        nu = C.c_int(); u = np.zeros(n, np.uint64)
        nv = emu.emu_win_extract(n, bx, by, fa, pa, 3, 40, 0, C.byref(nu), u)
        assert nu.value == len(ou) and np.array_equal(u[:nu.value], ou) and np.array_equal(bx[:nv], obx) and np.array_equal(by[:nv], oby), it
        n_cases += int(len(ou) != len(ou0) or not np.array_equal(ou, ou0))
    assert n_cases >= 3        # (the cDNA cost changes the chains in a good part of the cases: the test would not pass with the genomic cost)
