"""Kernel logic on the CPU: the product kernel headers compiled against the host wavefront emulator
(tests/simt_emu) must agree bit-for-bit with the oracle. This is how kernel bugs are found without a GPU."""
import collections
import ctypes as C
import numpy as np
import pytest
import wmtest as W
import kswcases
from winnowmap_amd import build


def _load_emu(defines=()):
    E = C.CDLL(build.build_emu(defines))
    E.emu_ksw_extd2.argtypes = [C.c_int, W.u8p, C.c_int, W.u8p, W.i8p] + [C.c_int] * 9 + [W.i32p, W.u32p, C.c_int, C.POINTER(C.c_int)]
    E.emu_sketch.argtypes = [C.c_int, W.u8p, W.u64p, W.i32p, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, W.u64p, W.u64p, W.u64p, W.i32p, W.i32p]
    E.emu_sketch_coop.argtypes = E.emu_sketch.argtypes
    E.emu_seed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, W.u64p, W.u64p, C.c_int, C.c_int, C.c_int, C.c_int, W.u64p, W.u64p, C.c_int, W.i32p]
    E.emu_chain_fill.argtypes = [C.c_int64, W.u64p, W.u64p] + [C.c_int] * 6 + [C.c_float, C.c_float, W.i32p, W.i32p, W.i32p]
    return E


@pytest.fixture(scope="module")
def emu():
    return _load_emu()


# kernel variants next to the library default (winnowmap_amd/build.py WM_KERNEL_DEFINES): same tests, same bar. The default build has
# WM_KSW_ROR=1 since round 3; "readlane" is the former default (neighbour values through v_readlane + scalar fill)
KSW_VARIANTS = {"default": (), "readlane": ("WM_KSW_ROR=0",)}


@pytest.fixture(scope="module", params=sorted(KSW_VARIANTS))
def emu_v(request):
    return _load_emu(KSW_VARIANTS[request.param])


def emu_ksw(E, c, force=-1):
    ez = np.zeros(10, np.int32)
    cig = np.zeros(len(c["q"]) + len(c["t"]) + 4, np.uint32)
    k = C.c_int()
    n = E.emu_ksw_extd2(len(c["q"]), c["q"], len(c["t"]), c["t"], W.simple_mat(c["a"], c["b"], 1), c["q_"], c["e"], c["q2"], c["e2"],
                        c["w"], c["zdrop"], c["end_bonus"], c["flag"], force, ez, cig, len(cig), C.byref(k))
    return n, ez, cig[:max(n, 0)], k.value


@pytest.mark.parametrize("seed", [1, 2])
def test_ksw_emulated_kernel_matches_oracle(emu_v, seed):
    emu = emu_v
    seen = set()
    for c in kswcases.make_cases(seed, 90, max_len=600):
        o = W.o_ksw_extd2(c["q"], c["t"], mat=W.simple_mat(c["a"], c["b"], 1), q=c["q_"], e=c["e"], q2=c["q2"], e2=c["e2"],
                          w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])
        # host's choice; the CLIP+HASN and the plain CLIP variant of every window size; the 2-pair test window (many re-bases / pair boundaries)
        for force in (-1, 3, 11, 19, 2, 10, 18, 0, 100, 102, 103):
            n, ez, cig, klass = emu_ksw(emu, c, force)
            if n < 0:
                continue
            seen.add(klass)
            assert [int(x) for x in ez] == [o[k] for k in W.EZ_FIELDS], (klass, c["flag"], c["w"])
            assert np.array_equal(cig, o["cigar"])
    assert len(seen) >= 10, seen


def test_block_ksw_kernels_match_oracle(emu):
    """The multi-wave kernels (LDS window 4096 / 8192, and the global-state variant incl. a small geometry that forces
    many chunks per row) on forced small cases and on natively wide bands."""
    from winnowmap_amd import synth
    cases = kswcases.make_cases(5, 24, max_len=400)
    rng = np.random.default_rng(9)
    for it in range(3):
        tl = int(rng.integers(1100, 1700))
        t = rng.integers(0, 4, tl).astype(np.uint8)
        q = synth.mutate_codes(t, rng, 0.04, 0.04, 0.05) if it else rng.integers(0, 4, 1200).astype(np.uint8)
        cases.append(dict(q=q, t=t, a=2, b=4, q_=4, e=2, q2=24, e2=1, w=[3001, 1500, 3001][it], zdrop=[400, 200, -1][it], end_bonus=-1, flag=[0x08, 0x40, 0xC2][it]))
    n_run = 0
    for c in cases:
        o = W.o_ksw_extd2(c["q"], c["t"], mat=W.simple_mat(c["a"], c["b"], 1), q=c["q_"], e=c["e"], q2=c["q2"], e2=c["e2"],
                          w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])
        for force in (24, 124, 25, 125, 26, 126, 16, 18, 19):
            n, ez, cig, klass = emu_ksw(emu, c, force)
            if n < 0:
                continue
            n_run += 1
            assert [int(x) for x in ez] == [o[k] for k in W.EZ_FIELDS], (force, len(c["q"]), len(c["t"]), c["flag"], c["w"])
            assert np.array_equal(cig, o["cigar"])
    assert n_run > 80


@pytest.fixture(scope="module")
def small_index():
    """a host-built index (product code, via the test harness) + its bloom filter, and windows of reads to sketch"""
    import tempfile
    from winnowmap_amd import synth
    H = C.CDLL(build.build_harness())
    H.h_index_build.restype = C.c_void_p
    H.h_index_build.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
    vp = C.POINTER(C.c_uint64)
    H.h_index_view.argtypes = [C.c_void_p, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    H.h_bloom_view.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint8))]
    H.h_chain_extract.restype = C.c_int64
    H.h_chain_extract.argtypes = [C.c_int64, W.u64p, W.u64p, W.i32p, W.i32p, W.i32p, C.c_int, C.c_int, C.POINTER(C.c_int), W.u64p, W.u64p, W.u64p]
    H.h_avg_qspan.restype = C.c_float
    H.h_avg_qspan.argtypes = [C.c_int64, W.u64p]
    H.h_index_get.argtypes = [C.c_void_p, C.c_uint64, W.u64p, C.c_int]
    d = tempfile.mkdtemp()
    ref = synth.make_reference(2, 200000, 3, repeat_frac=0.15)
    synth.write_fasta(d + "/ref.fa", ref)
    km, cnt = synth.repetitive_kmers(ref, 15)
    synth.write_kmer_list(d + "/rep.txt", km, cnt, 15)
    h = H.h_index_build((d + "/ref.fa").encode(), (d + "/rep.txt").encode(), 15, 50, 4)
    hk, hv, P = vp(), vp(), vp()
    hb = C.c_int(); ns = C.c_uint64(); npz = C.c_uint64()
    H.h_index_view(h, C.byref(hk), C.byref(hv), C.byref(P), C.byref(hb), C.byref(ns), C.byref(npz))
    tb = C.c_uint32(); salts = (C.c_uint32 * 2)(); bb = C.POINTER(C.c_uint8)()
    H.h_bloom_view(h, C.byref(tb), salts, C.byref(bb))
    reads, _ = synth.make_reads(ref, 6, 15000, 5)
    return dict(H=H, h=h, hk=hk, hv=hv, P=P, hbits=hb.value, tb=tb.value, salts=(salts[0], salts[1]), bloom_bits=bb, bloom=W.o_bloom(km), reads=reads, synth=synth)


@pytest.mark.parametrize("kernel,w,k", [("lane", 50, 15), ("coop", 50, 15), ("coop", 50, 19), ("coop", 100, 21), ("coop", 10, 15), ("coop", 5, 11), ("coop", 200, 27), ("lane", 20, 14)])
@pytest.mark.parametrize("packed", [0, 1])
def test_sketch_kernel_emulated_matches_oracle(emu, small_index, kernel, w, k, packed):
    """both sketch kernels: one lane per sequence (sketch_wave, any k) and one wavefront per sequence (sketch_coop, odd k); sequences as bytes, and as the
    mapper's resident reads are laid out in HBM (packed = 1: 2 bits per base + ambiguity bitmap, csrc/reads2bit.h — the k-mer of a position is one 64-bit window)"""
    S = small_index
    emu.emu_set_packed(packed)
    rng = np.random.default_rng(4)
    seqs = [r[st:st + 2000].copy() for r in S["reads"][:3] for st in range(0, 14000, 4000)]
    # masked reads as stage 2 sketches them (mapped stretches replaced by N, src/map.c:786-846), N runs of every length around w and k
    m = S["reads"][1].copy(); m[1000:6000] = 4; m[9000:9049] = 4; m[9100:9151] = 4; m[12000:12014] = 4; m[12100:12115] = 4
    seqs += [m, np.full(300, 4, np.uint8), np.concatenate([np.full(70, 4, np.uint8), S["reads"][2][:500], np.full(3, 4, np.uint8), S["reads"][2][500:520]])]
    for ln_ in (1, 14, 15, 16, 63, 64, 65, 66, 79, 113, 128, 129):       # lengths around k, w + k and the 64-position tile
        seqs.append(S["reads"][3][100:100 + ln_].copy())
    for it in range(12):                      # low-complexity / periodic sequences, some with N
        L, unit = int(rng.integers(100, 2500)), int(rng.integers(1, 13))
        s = np.tile(rng.integers(0, 4, unit), L // unit + 1)[:L].astype(np.uint8)
        s = S["synth"].mutate_codes(s, rng, 0.01, 0, 0)
        for _ in range(int(rng.integers(0, 3))):
            s[int(rng.integers(0, len(s)))] = 4
        seqs.append(s)
    seqs += [S["reads"][0], np.array([0, 1, 2], np.uint8)]
    lens = np.array([len(s) for s in seqs], np.int32)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    caps = (lens + 1).astype(np.int32)
    ooffs = np.concatenate([[0], np.cumsum(caps)[:-1]]).astype(np.uint64)
    ox = np.zeros(int(caps.sum()), np.uint64); oy = np.zeros(int(caps.sum()), np.uint64); counts = np.zeros(len(seqs), np.int32)
    fn = emu.emu_sketch if kernel == "lane" else emu.emu_sketch_coop
    try:
        fn(len(seqs), np.concatenate(seqs), offs, lens, w, k, S["tb"], S["salts"][0], S["salts"][1], C.cast(S["bloom_bits"], C.c_void_p), ox, oy, ooffs, caps, counts)
    finally:
        emu.emu_set_packed(0)
    for i, s in enumerate(seqs):
        ex, ey = W.o_sketch(bytes(s), w, k, rid=0, bloom=S["bloom"])
        n = counts[i]
        assert n == len(ex), (i, n, len(ex))
        assert np.array_equal(ox[int(ooffs[i]):int(ooffs[i]) + n], ex) and np.array_equal(oy[int(ooffs[i]):int(ooffs[i]) + n], ey)


@pytest.mark.parametrize("w,k,packed", [(50, 15, 0), (50, 15, 1), (10, 19, 1), (5, 11, 0), (100, 21, 1)])
def test_sketch_with_homopolymer_compression_matches_oracle(emu, small_index, w, k, packed):
    """MM_I_HPC (-H, src/sketch.c:152-163) in sketch_coop: the sequence is compacted into its runs (sketch_hpc_steps), the two phases run over the runs;
    a minimizer sits on the last base of its last run and carries the summed length of its k runs as span (none at 256 and beyond). Sequences with long
    runs, runs next to N and at both ends; bytes and packed reads. The oracle's HPC branch is pinned to the reference's mm_sketch (tests/test_oracle_vs_ref.py)."""
    S = small_index
    rng = np.random.default_rng(50 + k)
    seqs = [r[:6000].copy() for r in S["reads"][:3]]
    for it in range(20):
        L = int(rng.integers(20, 1500))
        runs = rng.integers(1, 7, L)
        runs[rng.integers(0, L, max(1, L // 40))] = rng.integers(15, 330, max(1, L // 40))
        base = rng.integers(0, 4, L)
        base[1:] = np.where(base[1:] == base[:-1], (base[1:] + 1) & 3, base[1:])
        s = np.repeat(base, runs).astype(np.uint8)
        for _ in range(int(rng.integers(0, 4))):
            p = int(rng.integers(0, len(s)))
            s[p:p + int(rng.integers(1, 4))] = 4
        seqs.append(s)
    seqs += [np.zeros(700, np.uint8), np.full(90, 4, np.uint8), np.array([2], np.uint8), np.array([1, 1, 1, 4, 4, 0, 0], np.uint8)]
    for ln_ in (63, 64, 65, 128, 129):
        seqs.append(S["reads"][3][200:200 + ln_].copy())
    lens = np.array([len(s) for s in seqs], np.int32)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    caps = (lens + 1).astype(np.int32)
    ooffs = np.concatenate([[0], np.cumsum(caps)[:-1]]).astype(np.uint64)
    ox = np.zeros(int(caps.sum()), np.uint64); oy = np.zeros(int(caps.sum()), np.uint64); counts = np.zeros(len(seqs), np.int32)
    emu.emu_set_packed(packed); emu.emu_set_hpc(1)
    try:
        emu.emu_sketch_coop(len(seqs), np.concatenate(seqs), offs, lens, w, k, S["tb"], S["salts"][0], S["salts"][1], C.cast(S["bloom_bits"], C.c_void_p), ox, oy, ooffs, caps, counts)
    finally:
        emu.emu_set_packed(0); emu.emu_set_hpc(0)
    n_span = 0
    for i, s in enumerate(seqs):
        ex, ey = W.o_sketch(bytes(s), w, k, rid=0, bloom=S["bloom"], hpc=True)
        n = counts[i]
        assert n == len(ex), (i, len(s), n, len(ex))
        assert np.array_equal(ox[int(ooffs[i]):int(ooffs[i]) + n], ex) and np.array_equal(oy[int(ooffs[i]):int(ooffs[i]) + n], ey), i
        n_span += int(np.count_nonzero((ex & np.uint64(0xff)) != np.uint64(k)))
    assert n_span > 100


@pytest.mark.parametrize("w,k,chunk", [(50, 15, 64), (50, 15, 257), (50, 19, 1000), (10, 15, 100), (100, 21, 333), (5, 11, 64), (200, 27, 512)])
@pytest.mark.parametrize("packed", [0, 1])
def test_sketch_in_chunks_matches_oracle(emu, small_index, w, k, chunk, packed):
    """sketch_p1_range / sketch_find_sync / sketch_p2_range (sketch_kernel.h): a sequence cut into chunks, one wavefront per chunk, as the sketch_long_*
    kernels run contigs and long reads. Chunks far shorter than in the product (64 .. 1 000 positions against 16 384) put a boundary into every
    kind of stretch: N runs, masked reads, homopolymers and short tandem repeats (equal orders: no sync position, the chunk is absorbed), sequence ends."""
    S = small_index
    emu.emu_sketch_chunked.argtypes = list(emu.emu_sketch.argtypes) + [C.c_int, C.POINTER(C.c_int32)]
    rng = np.random.default_rng(40 + chunk)
    seqs = [r.copy() for r in S["reads"][:2]]
    m = S["reads"][1].copy(); m[1000:6000] = 4; m[9000:9049] = 4; m[9100:9151] = 4; m[12000:12014] = 4; m[12100:12115] = 4
    seqs += [m, np.full(300, 4, np.uint8), np.concatenate([np.full(70, 4, np.uint8), S["reads"][2][:500], np.full(3, 4, np.uint8), S["reads"][2][500:520]])]
    for ln_ in (1, 14, 15, 16, 63, 64, 65, 66, 79, 113, 128, 129, chunk - 1, chunk, chunk + 1, 2 * chunk, 2 * chunk + k):
        seqs.append(S["reads"][3][100:100 + ln_].copy())
    for it in range(16):                      # low-complexity / periodic sequences (ties over whole chunks), some with N; and such stretches inside random sequence
        L, unit = int(rng.integers(100, 4000)), int(rng.integers(1, 13))
        s = np.tile(rng.integers(0, 4, unit), L // unit + 1)[:L].astype(np.uint8)
        if it % 2:
            s = S["synth"].mutate_codes(s, rng, 0.01, 0, 0)
        for _ in range(int(rng.integers(0, 3))):
            s[int(rng.integers(0, len(s)))] = 4
        if it % 4 == 0:
            s = np.concatenate([S["reads"][4][:int(rng.integers(50, 900))], s, S["reads"][4][2000:2000 + int(rng.integers(50, 900))]])
        seqs.append(s)
    seqs.append(np.zeros(3 * chunk + 17, np.uint8))                    # one homopolymer across several chunks
    lens = np.array([len(s) for s in seqs], np.int32)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    caps = (lens + 1).astype(np.int32)
    ooffs = np.concatenate([[0], np.cumsum(caps)[:-1]]).astype(np.uint64)
    ox = np.zeros(int(caps.sum()), np.uint64); oy = np.zeros(int(caps.sum()), np.uint64); counts = np.zeros(len(seqs), np.int32)
    absorbed = C.c_int32()
    emu.emu_set_packed(packed)
    try:
        rc = emu.emu_sketch_chunked(len(seqs), np.concatenate(seqs), offs, lens, w, k, S["tb"], S["salts"][0], S["salts"][1], C.cast(S["bloom_bits"], C.c_void_p), ox, oy, ooffs, caps, counts,
                                    chunk, C.byref(absorbed))
    finally:
        emu.emu_set_packed(0)
    assert rc == 0
    for i, s in enumerate(seqs):
        ex, ey = W.o_sketch(bytes(s), w, k, rid=0, bloom=S["bloom"])
        n = counts[i]
        assert n == len(ex), (i, len(s), n, len(ex))
        assert np.array_equal(ox[int(ooffs[i]):int(ooffs[i]) + n], ex) and np.array_equal(oy[int(ooffs[i]):int(ooffs[i]) + n], ey), (i, len(s))
    assert absorbed.value > 0          # chunks without a sync position occurred (N runs, homopolymers)


def _expected_anchors(S, mx, my, qlen, max_occ=5000):
    """collect_seed_hits restated in numpy/python on the product index (src/map.c:97-130,222-254)"""
    H, h = S["H"], S["h"]
    ex, ey = [], []
    buf = np.zeros(6000, np.uint64)
    rep_st = rep_en = rep = 0
    for i, (x, y) in enumerate(zip(mx, my)):
        t = H.h_index_get(h, int(x) >> 8, buf, 6000)
        qpos = int(y) & 0xffffffff; span = int(x) & 0xff
        if t >= max_occ:
            en = (qpos >> 1) + 1; st = en - span
            if st > rep_en:
                rep += rep_en - rep_st; rep_st, rep_en = st, en
            else:
                rep_en = en
            continue
        tand = (i > 0 and int(mx[i - 1]) >> 8 == int(x) >> 8) or (i < len(mx) - 1 and int(mx[i + 1]) >> 8 == int(x) >> 8)
        for r in buf[:t]:
            r = int(r); rpos = (r & 0xffffffff) >> 1
            if (r & 1) == (qpos & 1):
                X = (r & 0xffffffff00000000) | rpos; Y = span << 32 | qpos >> 1
            else:
                X = 1 << 63 | (r & 0xffffffff00000000) | rpos; Y = span << 32 | (qlen - ((qpos >> 1) + 1 - span) - 1)
            if tand:
                Y |= 1 << 42
            ex.append(X); ey.append(Y)
    rep += rep_en - rep_st
    return np.array(ex, np.uint64), np.array(ey, np.uint64), rep


def test_seed_and_chain_kernels_emulated_match_oracle(emu, small_index):
    S = small_index
    H = S["H"]
    windows = [r[st:st + 2000].copy() for r in S["reads"][:2] for st in range(0, 14000, 3500)] + [S["reads"][2], S["reads"][3]]
    n_chain = 0
    for wi, s in enumerate(windows):
        mx, my = W.o_sketch(bytes(s), 50, 15, bloom=S["bloom"])
        cap = 200000
        ax = np.zeros(cap, np.uint64); ay = np.zeros(cap, np.uint64); res = np.zeros(2, np.int32)
        emu.emu_seed(C.cast(S["hk"], C.c_void_p), C.cast(S["hv"], C.c_void_p), C.cast(S["P"], C.c_void_p), S["hbits"], mx, my, len(mx), len(s), 5000, 0, ax, ay, cap, res)
        n = res[0]
        ax, ay = ax[:n], ay[:n]
        ex, ey, rep = _expected_anchors(S, mx, my, len(s))
        assert np.array_equal(ax, ex) and np.array_equal(ay, ey) and res[1] == rep
        if n == 0:
            continue
        sx, sy = W.o_radix_sort_128x(ax, ay)
        avg = H.h_avg_qspan(n, sy)
        prm = [dict(max_dist_x=5000, min_dist_x=1000, max_dist_y=5000, bw=500), dict(max_dist_x=16000, min_dist_x=1000, max_dist_y=16000, bw=2000)][wi % 2]
        ou, obx, oby = W.o_chain_dp(sx, sy, **prm)
        # (window, waves, tiles per wave and step): single wave in-window; single wave with a 128-anchor window (wrap + global fallback); 4 and 8 cooperating
        # waves (chain_block); chain_block_wide with 2 x 2, 4 x 3 and 16 x 5 tiles per step
        for win, nwv, kt in ((0, 0, 0), (128, 0, 0), (128, 4, 0), (0, 8, 0), (128, 2, 2), (0, 4, 3), (0, 16, 5)):
            fa = np.zeros(n, np.int32); pa = np.zeros(n, np.int32); va = np.zeros(n, np.int32)
            emu.emu_chain_fill(n, sx, sy, prm["max_dist_x"], prm["min_dist_x"], prm["max_dist_y"], prm["bw"], 25 | (win << 16), 5000 | (nwv << 24) | (kt << 20), avg, 1.0, fa, pa, va)
            u = np.zeros(n, np.uint64); bx = np.zeros(n, np.uint64); by = np.zeros(n, np.uint64); nu = C.c_int()
            nv = H.h_chain_extract(n, sx, sy, fa, pa, va, 3, 40, C.byref(nu), u, bx, by)
            assert np.array_equal(u[:nu.value], ou) and np.array_equal(bx[:nv], obx) and np.array_equal(by[:nv], oby), (wi, n, win, nwv, kt)
            n_chain += 1
    assert n_chain >= 40


def _satellite_anchors(seed, n_mini, copies, period=171, max_iter_hint=0):
    """anchors of a query that crosses a tandem array: every query minimizer hits `copies` neighbouring copies of the monomer (dense, long predecessor scans)"""
    rng = np.random.default_rng(seed)
    qpos = np.cumsum(rng.integers(5, 40, n_mini)).astype(np.int64) + 100
    xs, ys = [], []
    for q in qpos:
        for k in rng.choice(np.arange(-copies, copies + 1), size=int(rng.integers(max(1, copies // 2), 2 * copies + 1)), replace=False):
            r = 1_000_000 + int(q) + int(k) * period + int(rng.integers(-2, 3))
            xs.append(r); ys.append((15 << 32) | int(q))
    return W.o_radix_sort_128x(np.array(xs, np.uint64), np.array(ys, np.uint64))


def test_chain_fill_over_the_whole_predecessor_window_matches_oracle(emu):
    """chain_block_wide (seedchain_kernel.h): every wavefront scores KT tiles of the predecessor window per step — on satellite-like anchor sets (hundreds of
    predecessors per anchor, marks everywhere, max_skip breaks in the middle of a window, max_iter clipping the window) against the oracle's mm_chain_dp fill
    (f and p through the chains they give). Geometries from 2 x 2 tiles per step (many steps per anchor, the carried automaton state) to 16 x 5 (one step)."""
    H = C.CDLL(build.build_harness())
    H.h_chain_extract.restype = C.c_int64
    H.h_chain_extract.argtypes = [C.c_int64, W.u64p, W.u64p, W.i32p, W.i32p, W.i32p, C.c_int, C.c_int, C.POINTER(C.c_int), W.u64p, W.u64p, W.u64p]
    H.h_avg_qspan.restype = C.c_float
    H.h_avg_qspan.argtypes = [C.c_int64, W.u64p]
    n_run = 0
    for seed, n_mini, copies, max_iter, max_skip in ((1, 60, 12, 5000, 25), (2, 40, 30, 5000, 25), (3, 50, 20, 300, 25), (4, 30, 40, 5000, 3), (5, 25, 60, 700, 40)):
        sx, sy = _satellite_anchors(seed, n_mini, copies)
        n = len(sx)
        avg = H.h_avg_qspan(n, sy)
        prm = dict(max_dist_x=16000, min_dist_x=1000, max_dist_y=16000, bw=2000)
        ou, obx, oby = W.o_chain_dp(sx, sy, max_skip=max_skip, max_iter=max_iter, **prm)
        # (LDS window, wavefronts, tiles per wavefront and step, tiles per wavefront in an anchor's FIRST step: 0 = as in the others, the production setting is 1)
        for win, nwv, kt, ktf in ((0, 8, 0, 0), (128, 2, 2, 0), (256, 4, 1, 0), (0, 4, 3, 0), (0, 16, 5, 0), (512, 16, 5, 0),
                                  (128, 2, 2, 1), (0, 4, 3, 1), (0, 16, 5, 1), (512, 16, 5, 1), (256, 8, 5, 2), (0, 2, 10, 3)):
            fa = np.zeros(n, np.int32); pa = np.zeros(n, np.int32); va = np.zeros(n, np.int32)
            emu.emu_chain_fill(n, sx, sy, prm["max_dist_x"], prm["min_dist_x"], prm["max_dist_y"], prm["bw"], max_skip | (ktf << 8) | (win << 16), max_iter | (nwv << 24) | (kt << 20), avg, 1.0, fa, pa, va)
            u = np.zeros(n, np.uint64); bx = np.zeros(n, np.uint64); by = np.zeros(n, np.uint64); nu = C.c_int()
            nv = H.h_chain_extract(n, sx, sy, fa, pa, va, 3, 40, C.byref(nu), u, bx, by)
            assert np.array_equal(u[:nu.value], ou) and np.array_equal(bx[:nv], obx) and np.array_equal(by[:nv], oby), (seed, n, win, nwv, kt, ktf)
            n_run += 1
    assert n_run == 60


def test_packed_multiwave_ksw_kernel_matches_oracle(emu_v):
    """ksw_dp_pmulti (ksw_packed_multi_kernel.h): the packed two-cells-per-lane machine over several wavefronts. Small geometries (2 waves x 1
    pair = 256 lanes, 3 waves x 2 pairs = 768 lanes) force many re-bases, pair boundaries on every wavefront and hulls that sweep across the
    whole window on small random cases of every flag / band / scoring combination; the production geometries (4 x 4 pairs = 2048 lanes,
    4 x 8 = 4096 lanes) run on natively wide bands."""
    emu = emu_v
    from winnowmap_amd import synth
    n_run = collections.Counter()
    for c in kswcases.make_cases(7, 80, max_len=500):
        o = W.o_ksw_extd2(c["q"], c["t"], mat=W.simple_mat(c["a"], c["b"], 1), q=c["q_"], e=c["e"], q2=c["q2"], e2=c["e2"],
                          w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])
        for force in (200, 202, 203, 210, 212, 213, 222):
            n, ez, cig, klass = emu_ksw(emu, c, force)
            if n < 0:
                continue
            n_run[force // 10] += 1
            assert [int(x) for x in ez] == [o[k] for k in W.EZ_FIELDS], (force, len(c["q"]), len(c["t"]), c["flag"], c["w"], c["zdrop"])
            assert np.array_equal(cig, o["cigar"]), (force, len(c["q"]), len(c["t"]), c["flag"], c["w"])
    # the band sitting on the last target lane when that lane starts a 16-lane group (en0 == st == tlen - 1 on rows that do not re-base):
    # H of the lane below the window is then the value the LAST re-base left (a bug of the first version of this kernel)
    rng = np.random.default_rng(5)
    for it, (ql, tl) in enumerate(((214, 209), (489, 193), (150, 129), (333, 321), (90, 65))):
        t = rng.integers(0, 4, tl).astype(np.uint8)
        q = np.concatenate([synth.mutate_codes(t, rng, 0.03, 0.02, 0.02), rng.integers(0, 4, ql).astype(np.uint8)])[:ql]
        c = dict(q=q, t=t, a=1, b=4, q_=6, e=2, q2=26, e2=1, w=[5, 5, 3, 9, 5][it], zdrop=[200, 400, -1, 100, 50][it], end_bonus=0, flag=[0x80, 0, 0x40, 0x80, 0][it])
        o = W.o_ksw_extd2(c["q"], c["t"], mat=W.simple_mat(c["a"], c["b"], 1), q=c["q_"], e=c["e"], q2=c["q2"], e2=c["e2"],
                          w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])
        for force in (2, 3, 202, 203, 212, 213, 223):
            n, ez, cig, klass = emu_ksw(emu, c, force)
            assert n >= 0, (force, ql, tl)
            assert [int(x) for x in ez] == [o[k] for k in W.EZ_FIELDS], (force, ql, tl, c["flag"], c["w"], c["zdrop"])
            assert np.array_equal(cig, o["cigar"]), (force, ql, tl)
    rng = np.random.default_rng(19)
    wide = []
    for it in range(4):
        tl = int(rng.integers(1300, 2200))
        t = rng.integers(0, 4, tl).astype(np.uint8)
        q = synth.mutate_codes(t, rng, 0.04, 0.04, 0.05) if it else rng.integers(0, 4, 1500).astype(np.uint8)
        if it == 3:
            q[700] = 4
        wide.append(dict(q=q, t=t, a=2, b=4, q_=4, e=2, q2=24, e2=1, w=[3001, 1200, 3001, 900][it], zdrop=[400, 200, -1, 400][it], end_bonus=[-1, 10, -1, 0][it], flag=[0x08, 0x40, 0xC2, 0x00][it]))
    for tl, fl, w in ((5200, 0x08, -1), (6100, 0x40, -1), (3900, 0x00, -1), (6000, 0x88, 4500)):       # hulls of 3900..6100 lanes: the 4096- and 8192-lane geometries
        t = rng.integers(0, 4, tl).astype(np.uint8)
        wide.append(dict(q=synth.mutate_codes(t, rng, 0.03, 0.03, 0.03), t=t, a=2, b=4, q_=4, e=2, q2=24, e2=1, w=w, zdrop=400, end_bonus=-1, flag=fl))
    for c in wide:
        o = W.o_ksw_extd2(c["q"], c["t"], mat=W.simple_mat(c["a"], c["b"], 1), q=c["q_"], e=c["e"], q2=c["q2"], e2=c["e2"],
                          w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])
        for force in (220, 222, 223, 230, 232, 233, 243, 253):          # (243 / 253: the CLIP + HASN instantiations of <4,8> and <8,8>, what WM_KSW_PMULTI launches)
            n, ez, cig, klass = emu_ksw(emu, c, force)
            if n < 0:
                continue
            n_run[force // 10] += 1
            assert [int(x) for x in ez] == [o[k] for k in W.EZ_FIELDS], (force, len(c["q"]), len(c["t"]), c["flag"], c["w"])
            assert np.array_equal(cig, o["cigar"]), (force, len(c["q"]), len(c["t"]), c["flag"], c["w"])
    assert n_run[20] > 30 and n_run[21] > 60 and n_run[22] >= 4 and n_run[23] >= 4 and n_run[24] >= 5 and n_run[25] >= 8, n_run


def emu_ksw_dual(E, nc, sc, jobs):
    """two alignments (or one) on ONE emulated wavefront (ksw_dual_kernel.h): rc, [(ez, cigar)]"""
    E.emu_ksw_dual.argtypes = [C.c_int, C.c_int, W.i32p, W.i32p, C.c_void_p, C.c_void_p, W.i32p, W.i32p, W.i8p] + [C.c_int] * 4 + [W.i32p, W.u32p, C.c_int, W.i32p]
    n = len(jobs)
    ql = np.array([len(j["q"]) for j in jobs] + [0] * (2 - n), np.int32)
    tl = np.array([len(j["t"]) for j in jobs] + [0] * (2 - n), np.int32)
    qs = (C.c_void_p * 2)(*[j["q"].ctypes.data for j in jobs] + [None] * (2 - n))
    ts = (C.c_void_p * 2)(*[j["t"].ctypes.data for j in jobs] + [None] * (2 - n))
    w = np.array([j["w"] for j in jobs] + [0] * (2 - n), np.int32)
    fl = np.array([j["flag"] for j in jobs] + [0] * (2 - n), np.int32)
    cap = int(max(ql + tl)) + 4
    ez = np.zeros(20, np.int32); cig = np.zeros(2 * cap, np.uint32); ncg = np.zeros(2, np.int32)
    rc = E.emu_ksw_dual(nc, int(n == 2), ql, tl, qs, ts, w, fl, W.simple_mat(sc["a"], sc["b"], 1), sc["q_"], sc["e"], sc["q2"], sc["e2"], ez, cig, cap, ncg)
    return rc, [(ez[10 * x:10 * x + 10].copy(), cig[x * cap:x * cap + ncg[x]].copy()) for x in range(n)]


def test_two_alignments_per_wavefront_kernel_matches_oracle(emu):
    """ksw_dp_dual (ksw_dual_kernel.h): two gap fills in the low / high halves of one wavefront's registers, every window size (4 / 8 / 16 chunks of 64 lanes;
    the product launches 8 and 16), against the oracle's ksw_extd2 one alignment at a time — ez fields and CIGAR (the traceback layout is shared with
    ksw_backtrack_thread). tools/dual_fuzz.py runs the same generator for as long as one likes."""
    n_al = n_single = n_mixed = 0
    for nc, sc, jobs in kswcases.dual_pairs(11, 260):
        rc, out = emu_ksw_dual(emu, nc, sc, jobs)
        if rc == -1:
            continue
        assert rc == 0, rc
        n_single += len(jobs) == 1
        n_mixed += len(jobs) == 2 and ((jobs[0]["flag"] ^ jobs[1]["flag"]) & 0x02) != 0
        for j, (ez, cig) in zip(jobs, out):
            o = W.o_ksw_extd2(j["q"], j["t"], mat=W.simple_mat(sc["a"], sc["b"], 1), q=sc["q_"], e=sc["e"], q2=sc["q2"], e2=sc["e2"], w=j["w"], zdrop=-1, end_bonus=0, flag=j["flag"])
            assert [int(v) for v in ez] == [o[k] for k in W.EZ_FIELDS], (nc, len(j["q"]), len(j["t"]), hex(j["flag"]), j["w"], [(len(a["q"]), len(a["t"])) for a in jobs])
            assert np.array_equal(cig, o["cigar"]), (nc, len(j["q"]), len(j["t"]), hex(j["flag"]), j["w"], [(len(a["q"]), len(a["t"])) for a in jobs])
            n_al += 1
    assert n_al >= 400 and n_single >= 20 and n_mixed >= 60, (n_al, n_single, n_mixed)


def test_exts2_splice_kernel_emulated_matches_oracle(emu):
    """ksw_dp_exts2 + ksw_exts2_backtrack_thread (ksw_exts2_kernel.h) against the oracle's ksw_exts2_sse restatement, which
    tests/test_oracle_vs_ref.py pins to the reference's own function: every splice flag, both gap alignments, exact / approximate maximum,
    extension-only, reversed operands, junction annotation, N bases, introns up to 2 kb (several 64-lane sweeps per row)."""
    emu.emu_ksw_exts2.argtypes = [C.c_int, W.u8p, C.c_int, W.u8p, W.i8p] + [C.c_int] * 7 + [C.c_void_p, W.i32p, W.u32p, C.c_int]
    n_intron = 0
    cases = kswcases.make_splice_cases(5, 160) + kswcases.make_splice_cases(6, 40, max_exon=300, max_intron=2000)
    for c in cases:
        o = W.o_ksw_exts2(c["q"], c["t"], mat=W.simple_mat(c["a"], c["b"], 1), q=c["q_"], e=c["e"], q2=c["q2"], noncan=c["noncan"], zdrop=c["zdrop"],
                          junc_bonus=c["junc_bonus"], flag=c["flag"], junc=c["junc"])
        ez = np.zeros(10, np.int32)
        cig = np.zeros(len(c["q"]) + len(c["t"]) + 4, np.uint32)
        jn = None if c["junc"] is None else np.ascontiguousarray(c["junc"], np.uint8)
        n = emu.emu_ksw_exts2(len(c["q"]), c["q"], len(c["t"]), c["t"], W.simple_mat(c["a"], c["b"], 1), c["q_"], c["e"], c["q2"], c["noncan"], c["zdrop"],
                              c["junc_bonus"], c["flag"], None if jn is None else jn.ctypes.data, ez, cig, len(cig))
        assert n >= 0
        assert [int(x) for x in ez] == [o[k] for k in W.EZ_FIELDS], (hex(c["flag"]), len(c["q"]), len(c["t"]), [int(x) for x in ez], [o[k] for k in W.EZ_FIELDS])
        assert np.array_equal(cig[:n], o["cigar"]), (hex(c["flag"]), W.cigar_str(cig[:n]), W.cigar_str(o["cigar"]))
        n_intron += any((int(x) & 0xf) == 3 for x in o["cigar"])
    assert n_intron > 60, n_intron


def test_wave_cooperative_backtrack_equals_the_single_thread_walk(emu):
    """ksw_backtrack_wave (one wavefront per alignment, traceback tiles through LDS) must produce the CIGAR of ksw_backtrack_thread, i.e. the
    oracle's: every flag / band / scoring combination, paths that leave the hull (forced states), re-fetches after 32 rows / 64 lanes."""
    from winnowmap_amd import synth
    emu.emu_set_coop_backtrack(1)
    try:
        cases = kswcases.make_cases(31, 120, max_len=700)
        rng = np.random.default_rng(4)
        for it in range(6):                     # long alignments with long gaps: many tiles, horizontal and vertical runs
            tl = int(rng.integers(1200, 2600))
            t = rng.integers(0, 4, tl).astype(np.uint8)
            q = synth.mutate_codes(t, rng, 0.05, 0.04, 0.05)
            if it % 2:
                q = np.concatenate([q[:400], q[400 + 150 * it:]])        # a long deletion
            else:
                q = np.concatenate([q[:500], rng.integers(0, 4, 90 * (it + 1)).astype(np.uint8), q[500:]])   # a long insertion
            cases.append(dict(q=q, t=t, a=2, b=4, q_=4, e=2, q2=24, e2=1, w=[3001, 900, -1][it % 3], zdrop=[400, -1, 200][it % 3], end_bonus=-1, flag=[0x08, 0x40, 0x82, 0x00, 0x48, 0x88][it]))
        for c in cases:
            o = W.o_ksw_extd2(c["q"], c["t"], mat=W.simple_mat(c["a"], c["b"], 1), q=c["q_"], e=c["e"], q2=c["q2"], e2=c["e2"],
                              w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])
            n, ez, cig, klass = emu_ksw(emu, c, -1)
            assert n >= 0
            assert np.array_equal(cig, o["cigar"]), (klass, len(c["q"]), len(c["t"]), hex(c["flag"]), c["w"], W.cigar_str(cig)[:80], W.cigar_str(o["cigar"])[:80])
    finally:
        emu.emu_set_coop_backtrack(0)


# ---- the stripe-pipelined ksw kernel (ksw_stripe_kernel.h): a library of its own, with event counters and a watchdog on the polling loops ----
STRIPE_EVENTS = {"stripe_switch": 0, "inject": 1, "ez_handover_at_activation": 2, "restart": 3, "edge_trk_handover": 4, "trk_handover": 5, "zdrop": 6,
                 "pri_eval": 7, "pri_skipped": 8, "early_message_hit": 9}


def _load_stripe(defines=()):
    E = C.CDLL(build.build_emu_stripe(defines))
    E.emu_stripe_extd2.argtypes = [C.c_int, W.u8p, C.c_int, W.u8p, W.i8p] + [C.c_int] * 9 + [W.i32p, W.u32p, C.c_int, C.POINTER(C.c_int)]
    E.emu_ksw_extd2 = E.emu_stripe_extd2
    return E


def _stripe_events(E):
    ev = (C.c_long * 16)()
    E.emu_stripe_events(ev)
    return {k: int(ev[i]) for k, i in STRIPE_EVENTS.items()}


def _stripe_run(E, cases, forces):
    n_run = collections.Counter()
    for c in cases:
        o = None
        for force in forces:
            n, ez, cig, klass = emu_ksw(E, c, force)
            if n < 0:
                continue            # the job does not fit this geometry, or needs CLIP / HASN and the variant lacks it
            if o is None:
                o = W.o_ksw_extd2(c["q"], c["t"], mat=W.simple_mat(c["a"], c["b"], 1), q=c["q_"], e=c["e"], q2=c["q2"], e2=c["e2"],
                                  w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])
            n_run[(force - 300) // 10] += 1
            assert [int(x) for x in ez] == [o[k] for k in W.EZ_FIELDS], (force, klass, len(c["q"]), len(c["t"]), c["w"], hex(c["flag"]), c["zdrop"])
            assert np.array_equal(cig, o["cigar"]), (force, klass, len(c["q"]), len(c["t"]), c["w"], hex(c["flag"]), c["zdrop"])
    return n_run


def test_stripe_pipelined_ksw_kernel_matches_oracle():
    """ksw_dp_stripe: wavefronts own fixed 128*BP-lane target stripes cyclically and hand (H, E, F, prefix maximum, bookkeeping state, the approximate
    maximum's track) to the right neighbour through row-stamped LDS messages. Small geometries (<1,2> = 2 waves x 128 lanes ... <2,3>) put stripe
    switches, hand-overs and ring wrap-arounds on small random cases of every flag / band / scoring combination; the product's geometries (<2,4>,
    <4,8>, <8,8>) run on natively wide bands. The host's thread scheduler supplies the interleavings; every rare path must have run."""
    E = _load_stripe()
    small = [300 + g * 10 + v for g in (0, 1, 7, 2) for v in (0, 2, 3)]
    n_run = _stripe_run(E, kswcases.stripe_edge_cases(3, 260, 700), small)
    n_run += _stripe_run(E, kswcases.stripe_cases(5, 60, 700), small)
    n_run += _stripe_run(E, kswcases.make_cases(9, 60, max_len=500), small)
    from winnowmap_amd import synth
    rng = np.random.default_rng(23)
    wide = []
    for tl, fl, w, zd in ((900, 0x08, -1, 400), (1500, 0x40, 700, 200), (1900, 0x00, -1, 400), (3300, 0x88, 1500, 400), (5200, 0x00, 2600, 100), (2500, 0x0A, -1, -1)):
        t = rng.integers(0, 4, tl).astype(np.uint8)
        q = synth.mutate_codes(t, rng, 0.03, 0.03, 0.03)
        if fl == 0x0A:
            q[len(q) // 2] = 4
        wide.append(dict(q=q, t=t, a=2, b=4, q_=4, e=2, q2=24, e2=1, w=w, zdrop=zd, end_bonus=-1, flag=fl))
    n_run += _stripe_run(E, wide, [300 + g * 10 + v for g in (3, 4, 5, 6) for v in (2, 3)])
    ev = _stripe_events(E)
    assert n_run[0] > 100 and n_run[1] > 150 and n_run[7] > 150 and n_run[2] > 200 and n_run[3] >= 2 and n_run[4] >= 4 and n_run[5] >= 8 and n_run[6] >= 10, n_run
    for k in ("stripe_switch", "inject", "ez_handover_at_activation", "edge_trk_handover", "trk_handover", "zdrop", "pri_eval", "pri_skipped"):
        assert ev[k] > 0, ev
    assert ev["restart"] == 0, ev              # with the real margin the safe-mode repeat is (provably) never needed


def test_stripe_sixteen_wavefront_geometries_match_oracle():
    """<1,16> and <2,16> (opt-in routing, wm_ksw_set_routing(2, ...)): the same kernel with one / two lane pairs per wavefront and sixteen wavefronts —
    small random cases (most stripes never reached, rings of 16) and hulls of 900 .. 3 700 lanes"""
    E = _load_stripe()
    forces = [300 + g * 10 + v for g in (8, 9) for v in (0, 2, 3)]
    n_run = _stripe_run(E, kswcases.stripe_edge_cases(13, 60, 700), forces)
    n_run += _stripe_run(E, kswcases.stripe_cases(15, 12, 1800), forces)
    from winnowmap_amd import synth
    rng = np.random.default_rng(29)
    wide = []
    for tl, fl, w, zd in ((950, 0x08, -1, 400), (1800, 0x40, 900, 200), (1850, 0x00, -1, 400), (2400, 0x88, 1200, 400), (3700, 0x00, -1, 100), (3000, 0x0A, -1, -1)):
        t = rng.integers(0, 4, tl).astype(np.uint8)
        q = synth.mutate_codes(t, rng, 0.03, 0.03, 0.03)
        if fl == 0x0A:
            q[len(q) // 2] = 4
        wide.append(dict(q=q, t=t, a=2, b=4, q_=4, e=2, q2=24, e2=1, w=w, zdrop=zd, end_bonus=-1, flag=fl))
    n_run += _stripe_run(E, wide, [300 + g * 10 + v for g in (8, 9) for v in (2, 3)])
    assert n_run[8] > 70 and n_run[9] > 70, n_run


def test_stripe_kernel_repeat_in_safe_mode():
    """the exact-maximum path skips the priority evaluation of a row whose prefix maximum cannot reach ez.max minus a margin; should a later stripe
    need a skipped priority after all, the job repeats with every priority evaluated. The real margin makes that unreachable; a useless margin
    (WM_STRIPE_TEST_SLACK) forces it, and the results must not change."""
    E = _load_stripe(("WM_STRIPE_TEST_SLACK=-100000",))
    forces = [300 + g * 10 + v for g in (0, 1, 7, 2) for v in (2, 3)]
    exact = [c for c in kswcases.stripe_edge_cases(4, 260, 700) + kswcases.stripe_cases(6, 40, 700) if not (c["flag"] & 0x08)]
    n_run = _stripe_run(E, exact, forces)
    ev = _stripe_events(E)
    assert sum(n_run.values()) > 300 and ev["restart"] > 20, (n_run, ev)


def test_stripe_kernel_with_the_left_message_loaded_before_the_cells():
    """WM_STRIPE_EARLY_MSG=1 (a build switch, off by default: ksw_stripe_kernel.h): the message of a row is read before the row's cells and used after
    them; when it was not there yet the polling loop takes over. Same results; both outcomes of the early read must have occurred."""
    E = _load_stripe(("WM_STRIPE_EARLY_MSG=1",))
    forces = [300 + g * 10 + v for g in (0, 1, 7, 2, 9) for v in (0, 2, 3)]
    n_run = _stripe_run(E, kswcases.stripe_edge_cases(31, 120, 700) + kswcases.stripe_cases(33, 30, 700), forces)
    ev = _stripe_events(E)
    assert sum(n_run.values()) > 800 and ev["early_message_hit"] > 1000 and ev["restart"] == 0, (n_run, ev)


def test_stripe_kernel_watchdog_gives_up_instead_of_hanging():
    """the kernel's polling loops have a budget; a wavefront that exceeds it stops the workgroup and the job comes back flagged (bt_i =
    KSW_BT_WATCHDOG -> emu_stripe_extd2 returns -4; on the device the traceback kernel raises the batch's error and wm_ksw_dev_run returns WM_EINTERNAL).
    A budget of a few polls expires in almost every job on the emulator (a neighbour is a host thread away); jobs that are not flagged are correct."""
    E = _load_stripe(("WM_STRIPE_SPIN_BUDGET=3",))
    cases = [c for c in kswcases.stripe_cases(21, 24, 700)]
    n_flagged = n_ok = 0
    for c in cases:
        for force in (300 + 1 * 10 + 3, 300 + 3 * 10 + 3):                 # <1,3> and <2,4>, CLIP + HASN instantiations (serve every job)
            n, ez, cig, klass = emu_ksw(E, c, force)
            if n == -4:
                n_flagged += 1
                continue
            if n < 0:
                continue
            o = W.o_ksw_extd2(c["q"], c["t"], mat=W.simple_mat(c["a"], c["b"], 1), q=c["q_"], e=c["e"], q2=c["q2"], e2=c["e2"],
                              w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])
            assert [int(x) for x in ez] == [o[k] for k in W.EZ_FIELDS] and np.array_equal(cig, o["cigar"])
            n_ok += 1
    assert n_flagged > 5, (n_flagged, n_ok)


# ---- the chained-workgroup ksw kernel (ksw_chain_kernel.h): every wavefront a workgroup of its own, row messages through a mailbox in global memory ----
CHAIN_EVENTS = {"stripe_switch": 0, "inject": 1, "ez_handover_at_activation": 2, "edge_trk_handover": 4, "trk_handover": 5, "zdrop": 6}


def _load_chain(defines=()):
    E = C.CDLL(build.build_emu_chain(defines))
    E.emu_chain_extd2.argtypes = [C.c_int, W.u8p, C.c_int, W.u8p, W.i8p] + [C.c_int] * 9 + [W.i32p, W.u32p, C.c_int, C.POINTER(C.c_int)]
    E.emu_ksw_extd2 = E.emu_chain_extd2
    return E


def _chain_events(E):
    ev = (C.c_long * 16)()
    E.emu_chain_events(ev)
    return {k: int(ev[i]) for k, i in CHAIN_EVENTS.items()}


def _chain_run(E, cases, forces):
    n_run = collections.Counter()
    for c in cases:
        o = None
        for force in forces:
            n, ez, cig, klass = emu_ksw(E, c, force)
            if n < 0:
                assert n == -1, (n, force, len(c["q"]), len(c["t"]), c["w"], hex(c["flag"]))      # -1: needs CLIP / HASN and the variant lacks it (or no such geometry); anything else is a failure
                continue
            if o is None:
                o = W.o_ksw_extd2(c["q"], c["t"], mat=W.simple_mat(c["a"], c["b"], 1), q=c["q_"], e=c["e"], q2=c["q2"], e2=c["e2"],
                                  w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])
            n_run[(force - 400) // 10] += 1
            assert [int(x) for x in ez] == [o[k] for k in W.EZ_FIELDS], (force, klass, len(c["q"]), len(c["t"]), c["w"], hex(c["flag"]), c["zdrop"])
            assert np.array_equal(cig, o["cigar"]), (force, klass, len(c["q"]), len(c["t"]), c["w"], hex(c["flag"]), c["zdrop"])
    return n_run
