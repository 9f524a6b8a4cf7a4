"""CPU: the walks over a finished alignment — mm_test_zdrop (src/align.c:32-89) and mm_update_extra with mm_fix_cigar (src/align.c:91-167, 240-286) —
as the mapper runs them (host/wm_align.cpp, csrc/cigar_walk.h: the same walk body the device's ksw_zdwalk_kernel compiles) against the REFERENCE's
own static functions (oracle/ref_align_shim.cpp: align.c compiled once more inside a namespace). VERDICT r4: these walks had only been compared
with their own source on another compiler."""
import ctypes as C
import numpy as np
import pytest
import wmtest as W
from winnowmap_amd import build, synth
from test_host_simd import random_alignment

pytestmark = pytest.mark.skipif(not W.have_ref(), reason="oracle/_ref not built")


@pytest.fixture(scope="module")
def LIBS():
    H = C.CDLL(build.build_harness())
    R = W.ref()
    i32p = np.ctypeslib.ndpointer(np.int32, flags="C")
    zargs = [C.c_int64] + [C.c_int] * 8 + [W.u8p, W.u8p]
    H.h_test_zdrop.argtypes = zargs + [W.u32p, C.c_int, W.i8p]
    R.refshim_test_zdrop.argtypes = zargs + [C.c_uint32, W.u32p, W.i8p]
    H.h_zdrop_walk.argtypes = [W.u8p, W.u8p, W.u32p, C.c_int] + [C.c_int] * 5 + [i32p]
    eargs = [C.c_int] * 5 + [W.u8p, W.u8p, W.i8p, C.c_int, C.c_int]
    H.h_update_extra.argtypes = eargs + [W.u32p, C.c_int, i32p, W.u32p, C.c_int, C.POINTER(C.c_int)]
    R.refshim_update_extra.argtypes = eargs + [C.c_uint32, W.u32p, i32p, W.u32p, C.c_int, C.POINTER(C.c_int)]
    return H, R


def test_zdrop_scan_and_verdict_equal_the_references_mm_test_zdrop(LIBS):
    H, R = LIBS
    rng = np.random.default_rng(31)
    n_inv = n_drop = 0
    for it in range(300):
        if it % 4 == 3:                      # an inverted stretch in the middle of an ungapped alignment: the inversion test (ksw_ll_i16 on the reverse complement) fires
            L = int(rng.integers(300, 900)); m = int(rng.integers(60, 200)); p = int(rng.integers(50, L - m - 50))
            t = rng.integers(0, 4, L).astype(np.uint8)
            q = t.copy(); q[p:p + m] = synth.revcomp_codes(t[p:p + m])
            q = np.where(rng.random(L) < 0.03, (q + 1) % 4, q).astype(np.uint8)
            cig = np.array([L << 4], np.uint32)
        else:
            cig, q, t = random_alignment(rng, int(rng.integers(1, 50)), 0.0 if it % 3 else 0.01)
            cig = cig[(cig & 0xf) != 3] if it % 2 else cig          # (with and without N operations)
            # rebuild the sequences' lengths for the filtered CIGAR: simply regenerate until consistent
            ql = sum(int(c >> 4) for c in cig if (c & 0xf) in (0, 1)); tl = sum(int(c >> 4) for c in cig if (c & 0xf) in (0, 2, 3))
            if ql > len(q) or tl > len(t):
                continue
        a, b, ambi = ((2, 4, 1), (1, 9, 2))[it % 2]
        mat = W.simple_mat(a, b, ambi)
        gq, ge = ((4, 2), (16, 2))[it % 2]
        # 1) the scan's maximum drop: binary search on the reference's verdict with the inversion test switched off (MM_F_FOR_ONLY)
        exp = np.zeros(5, np.int32)
        H.h_zdrop_walk(q, t, cig, len(cig), int(mat[0]), int(mat[1]), int(mat[24]), gq, ge, exp)
        lo, hi = -1, 1 << 22                  # verdict(T) = max_zdrop > T
        while hi - lo > 1:
            mid = (lo + hi) // 2
            v = R.refshim_test_zdrop(0x100000, mid, mid, gq, ge, 5000, 40, a, 80, q, t, len(cig), cig, mat)
            if v:
                lo = mid
            else:
                hi = mid
        assert int(exp[0]) == hi, (it, int(exp[0]), hi)
        n_drop += hi > 0
        # 2) the full verdict incl. the inversion test, at thresholds around the drop
        for zd, zdi in ((hi - 1, hi // 2), (hi + 5, hi // 3), (400, 200), (max(1, hi // 2), max(0, hi // 4))):
            args = (0, max(0, zd), max(0, zdi), gq, ge, 5000, 40, a, 80)
            want = R.refshim_test_zdrop(*args, q, t, len(cig), cig, mat)
            got = H.h_test_zdrop(*args, q, t, cig, len(cig), mat)
            assert got == want, (it, args, got, want)
            n_inv += want == 2
    assert n_drop > 150 and n_inv > 40, (n_drop, n_inv)


def test_update_extra_and_fix_cigar_equal_the_references(LIBS):
    H, R = LIBS
    rng = np.random.default_rng(32)
    n = 0
    for it in range(400):
        cig, q, t = random_alignment(rng, int(rng.integers(1, 60)), 0.0 if it % 3 else 0.02)
        if it % 5 == 0 and len(cig) > 2:      # a leading indel (mm_fix_cigar removes it and shifts the region's start), and runs of indels it merges
            cig = cig[1:]
            q = q[int(cig[0] >> 4) * 0:]      # (sequences are rebuilt below)
        ops = [(int(c) & 0xf, int(c) >> 4) for c in cig]
        ql = sum(l for o, l in ops if o in (0, 1)); tl = sum(l for o, l in ops if o in (0, 2, 3))
        # sequences of exactly the CIGAR's extent, with repeats around the indels so that left-alignment has something to do
        t = rng.integers(0, 2 if it % 4 == 0 else 4, tl).astype(np.uint8)
        q = np.zeros(ql, np.uint8); qi = ti = 0
        for o, l in ops:
            if o == 0:
                seg = t[ti:ti + l].copy(); mm = rng.random(l) < 0.05; seg[mm] = (seg[mm] + 1) % 4
                q[qi:qi + l] = seg; qi += l; ti += l
            elif o == 1:
                q[qi:qi + l] = rng.integers(0, 2 if it % 4 == 0 else 4, l); qi += l
            else:
                ti += l
        if it % 7 == 0 and ql > 10:
            q[int(rng.integers(0, ql))] = 4
        qp = np.concatenate([q, np.full(16, 4, np.uint8)]); tp = np.concatenate([t, np.full(16, 4, np.uint8)])
        a, b, ambi = ((2, 4, 1), (1, 9, 2))[it % 2]
        mat = W.simple_mat(a, b, ambi)
        gq, ge = ((4, 2), (16, 2))[it % 2]
        rev = it % 2
        o1, o2 = np.zeros(6, np.int32), np.zeros(6, np.int32)
        c1, c2 = np.zeros(len(cig) + 4, np.uint32), np.zeros(len(cig) + 4, np.uint32)
        n1, n2 = C.c_int(), C.c_int()
        R.refshim_update_extra(rev, 100, 100 + ql, 1000, 1000 + tl, qp, tp, mat, gq, ge, len(cig), cig, o1, c1, len(c1), C.byref(n1))
        rc = H.h_update_extra(rev, 100, 100 + ql, 1000, 1000 + tl, qp, tp, mat, gq, ge, cig, len(cig), o2, c2, len(c2), C.byref(n2))
        assert rc == 0 and n1.value == n2.value and np.array_equal(c1[:n1.value], c2[:n2.value]) and np.array_equal(o1, o2), (it, o1, o2, n1.value, n2.value)
        n += 1
    assert n == 400
