"""CPU: the vectorised host routines of the mapper (host/wm_align.cpp) against their portable specifications — ksw_ll_i16 on real 8 x int16
vectors vs the lane-by-lane form (which test_abi.py pins to the reference's ksw_ll_i16), and the 16-bases-per-step scan of mm_update_extra vs
wm_extra_walk (cigar_walk.h; the whole mapper is pinned to the reference end to end)."""
import ctypes as C
import numpy as np
import pytest
import wmtest as W
from winnowmap_amd import build, synth


@pytest.fixture(scope="module")
def H():
    L = C.CDLL(build.build_harness())
    i32p = np.ctypeslib.ndpointer(np.int32, flags="C")
    L.h_extra_walk_both.argtypes = [C.c_void_p, C.c_void_p, W.u32p, C.c_int] + [C.c_int] * 5 + [i32p, i32p]
    for f in (L.h_ll_i16, L.h_ll_i16_portable):
        f.restype = C.c_int
        f.argtypes = [C.c_int, W.u8p, C.c_int, W.u8p, W.i8p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return L


def random_alignment(rng, n_ops, n_frac):
    """a random CIGAR with the two sequences it aligns (0..4 codes), as the mapper's walks see them: runs of M with sparse mismatches / N, I, D, N ops"""
    ops, q, t = [], [], []
    for k in range(n_ops):
        op = 0 if k % 2 == 0 else int(rng.choice([1, 2, 2, 1, 3]))
        ln = int(rng.choice([1, 2, 5, 12, 15, 16, 17, 31, 32, 33, 70, 200])) if op == 0 else int(rng.integers(1, 40))
        ops.append(ln << 4 | op)
        if op == 0:
            a = rng.integers(0, 4, ln).astype(np.uint8)
            b = a.copy()
            mm = rng.random(ln) < 0.08
            b[mm] = (b[mm] + rng.integers(1, 4, int(mm.sum()))) % 4
            a[rng.random(ln) < n_frac] = 4
            b[rng.random(ln) < n_frac] = 4
            q.append(b); t.append(a)
        elif op == 1:
            x = rng.integers(0, 4, ln).astype(np.uint8); x[rng.random(ln) < n_frac] = 4; q.append(x)
        else:
            x = rng.integers(0, 4, ln).astype(np.uint8); x[rng.random(ln) < n_frac] = 4; t.append(x)
    return np.array(ops, np.uint32), np.concatenate(q), np.concatenate(t)


def test_update_extra_scan_16_bases_per_step_equals_the_generic_walk(H):
    rng = np.random.default_rng(5)
    for it in range(400):
        cig, q, t = random_alignment(rng, int(rng.integers(1, 60)), 0.0 if it % 3 else 0.02)
        # the fast walk may load 16 bytes beyond either sequence: poison the padding with values that would change the result if they were used
        qp = np.concatenate([q, np.full(16, it % 5, np.uint8)]); tp = np.concatenate([t, np.full(16, (it + 2) % 5, np.uint8)])
        a, b, ambi = ((2, -4, -1), (1, -9, -2), (5, -4, 0))[it % 3]
        go, ge = ((4, 2), (24, 1))[it % 2]
        f, g = np.zeros(6, np.int32), np.zeros(6, np.int32)
        H.h_extra_walk_both(qp.ctypes.data, tp.ctypes.data, cig, len(cig), a, b, ambi, go, ge, f, g)
        assert np.array_equal(f, g), (it, f, g)
        assert g[4] == len(q) and g[5] == len(t)


def test_ksw_ll_i16_on_vectors_equals_the_lane_by_lane_form(H):
    rng = np.random.default_rng(6)
    for it in range(300):
        q = rng.integers(0, 5 if it % 7 == 0 else 4, int(rng.integers(1, 700))).astype(np.uint8)
        t = rng.integers(0, 4, int(rng.integers(1, 700))).astype(np.uint8) if it % 2 == 0 else synth.mutate_codes(q, rng, 0.05, 0.05, 0.05)
        if len(t) == 0:
            t = q
        a, b = (1, 4) if it % 3 else (2, 4)
        mat = W.simple_mat(a, b, 1)
        go, ge = (4, 2) if it % 3 else (6, 1)
        r = []
        for f in (H.h_ll_i16, H.h_ll_i16_portable):
            qe, te = C.c_int(), C.c_int()
            s = f(len(q), q, len(t), t, mat, go, ge, C.byref(qe), C.byref(te))
            r.append((s, qe.value, te.value))
        assert r[0] == r[1], (it, r)


def test_usable_cores_follow_affinity_and_cgroup_quota():
    """wm::usable_cores (what wm_map_file_multi divides among its mapping calls) = the Python launcher's count (winnowmap_amd/dist.py)"""
    from winnowmap_amd import dist
    H = C.CDLL(build.build_harness())
    assert H.h_usable_cores() == dist.available_cores() >= 1


def test_read_codes_pack_to_two_bits_and_an_ambiguity_bitmap():
    """wm_pack_codes (csrc/reads2bit.h): base p = bits 2 * (p & 31) of pk[p >> 5], ambiguous iff bit p & 63 of nm[p >> 6] (its code bits 0); tail bits and
    the slack words zero. Every length around the 32- / 64-base word boundaries, N at random places, runs of N, all-N."""
    H = C.CDLL(build.build_harness())
    H.h_pk_words.restype = H.h_nm_words.restype = C.c_size_t
    H.h_pk_words.argtypes = H.h_nm_words.argtypes = [C.c_size_t]
    H.h_pack_codes.argtypes = [W.u8p, C.c_size_t, W.u64p, W.u64p]
    rng = np.random.default_rng(5)
    for n in [0, 1, 7, 8, 9, 31, 32, 33, 63, 64, 65, 127, 128, 129, 1000, 4096, 4097, 100003]:
        for kind in range(3):
            codes = rng.integers(0, 4, n).astype(np.uint8)
            if kind == 1 and n:
                codes[rng.integers(0, n, max(1, n // 9))] = 4
                codes[n // 3:n // 3 + 70] = 4
            if kind == 2:
                codes[:] = 4
            pkw, nmw = H.h_pk_words(n), H.h_nm_words(n)
            pk = np.full(pkw, 0xAAAAAAAAAAAAAAAA, np.uint64); nm = np.full(nmw, 0xFFFFFFFFFFFFFFFF, np.uint64)
            H.h_pack_codes(np.ascontiguousarray(codes), n, pk, nm)
            want_pk = np.zeros(pkw, np.uint64); want_nm = np.zeros(nmw, np.uint64)
            p = np.arange(n, dtype=np.uint64)
            c2 = np.where(codes >= 4, 0, codes).astype(np.uint64)
            np.bitwise_or.at(want_pk, (p >> np.uint64(5)).astype(np.int64), c2 << (np.uint64(2) * (p & np.uint64(31))))
            np.bitwise_or.at(want_nm, (p >> np.uint64(6)).astype(np.int64), (codes >= 4).astype(np.uint64) << (p & np.uint64(63)))
            assert np.array_equal(pk, want_pk) and np.array_equal(nm, want_nm), (n, kind)
