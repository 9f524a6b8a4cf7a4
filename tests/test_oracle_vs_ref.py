"""Pins the oracle: (1) known-answer vectors recorded from the compiled reference,
(2) differential tests against the REAL reference library when oracle/_ref is available (built from
/root/reference in the build container; prebuilt on the GPU box), (3) committed golden fixtures."""
import os
import struct
import tempfile
import numpy as np
import pytest
import wmtest as W
import kswcases
from winnowmap_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _need_ref():
    if not W.have_ref() and os.path.exists("/root/reference/src/map.c"):
        from winnowmap_amd import build
        build.build_oracle()
    if not W.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference here)")


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def test_appendix_a_hashes_and_bloom():
    O = W.oracle()
    m30 = (1 << 30) - 1
    assert [O.wmo_hash64(k, m30) for k in (0, 1, 0x2aaaaaaa, 0x12345678)] == [0x3ff06f15, 0x3794f8e6, 0x304a3cb6, 0x173acecc]
    assert [O.wmo_fmix64(k) for k in (0, 1, 0x2aaaaaaa, 0x12345678)] == [0, 0xb456bcfc34c2cb2c, 0xc4a4eac8c981a727, 0xd930745910885960]
    assert bits(-O.wmo_order(1, 0)) == 0x3fe68ad79f869859 and bits(-O.wmo_order(1, 1)) == 0x3faf0cd842d1401f
    assert bits(-O.wmo_order(0x12345678, 0)) == 0x3feb260e8b22110b and bits(-O.wmo_order(0x12345678, 1)) == 0x3fd12d8392e5f572
    assert O.wmo_bloom_hash(0x0000000123456789, 0x28415a75) == 0xabd7a366
    assert O.wmo_bloom_hash(0x0000000123456789, 0xa90f1fdc) == 0x1abba88f
    for n, bits_ in ((1000, 14384), (100000, 1437768), (500000, 7188824), (2000000, 28755280)):
        f = O.wmo_bloom_new(n)
        tb, salts, _ = W.o_bloom_view(f)
        assert tb == bits_ and salts == (0x28415a75, 0xa90f1fdc)
        O.wmo_bloom_free(f)


def test_appendix_a_sketch():
    x, y = W.o_sketch("ACGTTGCATGCCGATAGGCTTAACGGATCGATTTACGCGATATATATATCGGCTAGCTAGGATCCGAT", 5, 7)
    got = [(int(a) >> 8, (int(b) & 0xffffffff) >> 1, int(b) & 1) for a, b in zip(x, y)]
    exp = [(0x1eef, 7, 0), (0x2685, 12, 0), (0x1b4b, 13, 0), (0x2d59, 16, 0), (0xb7, 21, 0), (0xb13, 25, 1), (0x32a, 28, 0), (0x645, 29, 1),
           (0x12cb, 33, 1), (0x715, 36, 0), (0x142c, 40, 0), (0x210e, 42, 1), (0x1fa7, 43, 0), (0x3ba7, 48, 1), (0x1fa7, 50, 1), (0x1d2c, 53, 1),
           (0x3f17, 58, 0), (0x3f17, 59, 1), (0x1e21, 60, 0), (0x357, 63, 1), (0x1a4b, 66, 0)]
    assert got == exp and all(int(a) & 0xff == 7 for a in x)


def test_appendix_a_ksw():
    q = W.enc("ACGTTGCATGCCGATAGGCTTAACGGATCGATTTACGCG")
    t = W.enc("ACGTTGCATGCGATAGGCTTAACGGTTATCGATTTACGAG")
    o = W.o_ksw_extd2(q, t, flag=0x08)
    assert (o["score"], o["max"], o["max_q"], o["max_t"], o["mqe"], W.cigar_str(o["cigar"])) == (56, 0, -1, -1, -0x40000000, "10M1I15M2D13M")
    o = W.o_ksw_extd2(q, t, flag=0x40)
    assert (o["score"], o["max"], o["max_q"], o["max_t"], o["mqe"], o["mqe_t"], W.cigar_str(o["cigar"])) == (56, 58, 36, 37, 56, 39, "10M1I15M2D11M")
    o = W.o_ksw_extd2(q, t, flag=0xC2)
    assert (o["score"], o["max"], W.cigar_str(o["cigar"])) == (56, 58, "11M2D14M1I11M")
    o = W.o_ksw_extd2(q, t, q=4, e=2, q2=4, e2=2, flag=0)      # single-affine through the dual-affine machine
    assert (o["score"], o["max"], W.cigar_str(o["cigar"])) == (56, 58, "10M1I15M2D13M")


def test_ksw_vs_reference():
    _need_ref()
    for c in kswcases.make_cases(11, 300, max_len=900):
        kw = dict(mat=W.simple_mat(c["a"], c["b"], 1), q=c["q_"], e=c["e"], q2=c["q2"], e2=c["e2"], w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])
        o, r = W.o_ksw_extd2(c["q"], c["t"], **kw), W.r_ksw_extd2(c["q"], c["t"], **kw)
        assert all(o[k] == r[k] for k in W.EZ_FIELDS) and np.array_equal(o["cigar"], r["cigar"])


def test_extz2_equals_extd2_with_equal_pieces():
    _need_ref()
    import ctypes as C
    for c in kswcases.make_cases(12, 120, max_len=700, preset=2):
        mat = W.simple_mat(c["a"], c["b"], 1)
        ez = np.zeros(10, np.int32)
        cig = np.zeros(len(c["q"]) + len(c["t"]) + 4, np.uint32)
        n = W.ref().refshim_ksw_extz2(len(c["q"]), c["q"], len(c["t"]), c["t"], mat, c["q_"], c["e"], c["w"], c["zdrop"], c["end_bonus"], c["flag"], ez, cig, len(cig))
        o = W.o_ksw_extd2(c["q"], c["t"], mat=mat, q=c["q_"], e=c["e"], q2=c["q_"], e2=c["e"], w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])
        # mte/mte_q are never read by the mapper and differ in the offset formulation; everything the caller uses must agree
        for i, k in enumerate(W.EZ_FIELDS):
            if k in ("mte", "mte_q"):
                continue
            assert o[k] == int(ez[i]), (k, o[k], int(ez[i]))
        assert np.array_equal(o["cigar"], cig[:n])


def test_sort_ll_vs_reference():
    _need_ref()
    rng = np.random.default_rng(5)
    for n in (0, 1, 2, 63, 64, 65, 200, 5000, 20000):
        for mode in range(3):
            x = rng.integers(0, 2 ** 63, n, dtype=np.uint64) if mode == 0 else rng.integers(0, 50, n, dtype=np.uint64) if mode == 1 \
                else (rng.integers(0, 2, n, dtype=np.uint64) << np.uint64(63)) | rng.integers(0, 3000, n, dtype=np.uint64)
            y = np.arange(n, dtype=np.uint64)
            ox, oy = W.o_radix_sort_128x(x, y)
            rx, ry = W.r_radix_sort_128x(x, y)
            assert np.array_equal(ox, rx) and np.array_equal(oy, ry)
    for it in range(120):
        q = rng.integers(0, 4, int(rng.integers(1, 400))).astype(np.uint8)
        t = rng.integers(0, 4, int(rng.integers(1, 400))).astype(np.uint8) if it % 2 == 0 else synth.mutate_codes(q, rng, 0.05, 0.05, 0.05)
        if len(t) == 0:
            t = q
        assert W.o_ksw_ll(q, t) == W.r_ksw_ll(q, t)


@pytest.fixture(scope="module")
def ref_index():
    _need_ref()
    d = tempfile.mkdtemp()
    ref = synth.make_reference(2, 300000, 3, repeat_frac=0.1)
    synth.write_fasta(d + "/ref.fa", ref)
    km, cnt = synth.repetitive_kmers(ref, 15)
    synth.write_kmer_list(d + "/rep.txt", km, cnt, 15)
    mi = W.ref().refshim_idx_build((d + "/ref.fa").encode(), (d + "/rep.txt").encode(), 15, 50, 2)
    return ref, km, mi


def test_bloom_and_sketch_vs_reference(ref_index):
    ref, km, mi = ref_index
    f = W.o_bloom(km)
    tb, salts, bits_ = W.o_bloom_view(f)
    rb = np.zeros(tb // 8, np.uint8)
    assert W.ref().refshim_bloom_table_bits(mi) == tb
    W.ref().refshim_bloom_table_bytes(mi, rb.ctypes.data, len(rb))
    assert np.array_equal(bits_, rb)
    reads, _ = synth.make_reads(ref, 12, 3000, 7)
    rng = np.random.default_rng(9)
    seqs = [synth.codes_to_ascii(r) for r in reads] + [synth.codes_to_ascii(ref[0][:40000])]
    for it in range(40):      # adversarial: short-period tandem repeats, N runs, lower case
        L, unit = int(rng.integers(300, 4000)), int(rng.integers(1, 13))
        s = np.tile(rng.integers(0, 4, unit), L // unit + 1)[:L].astype(np.uint8)
        a = bytearray(synth.codes_to_ascii(synth.mutate_codes(s, rng, 0.01, 0.0, 0.0)))
        for _ in range(int(rng.integers(0, 4))):
            p = int(rng.integers(0, len(a) - 20))
            a[p:p + int(rng.integers(1, 20))] = b"N"
        seqs.append(bytes(a).lower() if it % 3 == 0 else bytes(a))
    for s in seqs:
        for w, k in ((50, 15), (10, 15), (5, 7)):
            ox, oy = W.o_sketch(s, w, k, rid=3, bloom=f)
            rx, ry = W.r_sketch(mi, s, w, k, rid=3)
            assert np.array_equal(ox, rx) and np.array_equal(oy, ry)
    # homopolymer compression (-H, src/sketch.c:152-163): runs of every length incl. beyond the 255-base span limit, runs next to N, at the sequence ends
    for it in range(30):
        L = int(rng.integers(200, 3000))
        runs = rng.integers(1, 9, L)
        runs[rng.integers(0, L, L // 50)] = rng.integers(20, 400, L // 50)
        base = rng.integers(0, 4, L)
        base[1:] = np.where(base[1:] == base[:-1], (base[1:] + 1) & 3, base[1:])
        a = bytearray(synth.codes_to_ascii(np.repeat(base, runs).astype(np.uint8)))
        for _ in range(int(rng.integers(0, 5))):
            p = int(rng.integers(0, len(a) - 3))
            a[p:p + int(rng.integers(1, 4))] = b"N"
        seqs.append(bytes(a))
    n_hpc = 0
    for s in seqs:
        for w, k in ((50, 15), (10, 19), (5, 7), (10, 14)):
            ox, oy = W.o_sketch(s, w, k, rid=3, bloom=f, hpc=True)
            rx, ry = W.r_sketch(mi, s, w, k, rid=3, hpc=True)
            assert np.array_equal(ox, rx) and np.array_equal(oy, ry)
            n_hpc += int(np.count_nonzero((ox & np.uint64(0xff)) != np.uint64(k)))
    assert n_hpc > 1000                        # spans other than k occurred


def ref_anchors(mi, seq, w, k, mid_occ=5000):
    """collect_seed_hits (src/map.c:222-254) composed from the reference's own mm_sketch / mm_idx_get."""
    mx, my = W.r_sketch(mi, seq, w, k)
    qlen = len(seq)
    ax, ay = [], []
    buf = np.zeros(6000, np.uint64)
    for x, y in zip(mx, my):
        n = W.ref().refshim_idx_get(mi, int(x) >> 8, buf, 6000)
        if n >= mid_occ:
            continue
        qpos, span = int(y) & 0xffffffff, int(x) & 0xff
        for r in buf[:n]:
            r = int(r)
            rpos = (r & 0xffffffff) >> 1
            if (r & 1) == (qpos & 1):
                ax.append((r & 0xffffffff00000000) | rpos); ay.append(span << 32 | qpos >> 1)
            else:
                ax.append(1 << 63 | (r & 0xffffffff00000000) | rpos); ay.append(span << 32 | (qlen - ((qpos >> 1) + 1 - span) - 1))
    return W.r_radix_sort_128x(np.array(ax, np.uint64), np.array(ay, np.uint64))


def test_chain_vs_reference(ref_index):
    ref, km, mi = ref_index
    reads, _ = synth.make_reads(ref, 14, 3000, 7)
    reads += synth.make_reads(ref, 6, 15000, 9)[0]
    for r in reads:
        ax, ay = ref_anchors(mi, synth.codes_to_ascii(r), 50, 15)
        # (the last: splice mode — cDNA gap cost of src/chain.c:69-74 with the preset's intron-sized distances, src/options.c:122)
        for prm in (dict(), dict(max_dist_x=16000, max_dist_y=16000, bw=2000), dict(max_dist_x=200000, max_dist_y=2000, bw=200000, is_cdna=1)):
            ou, obx, oby = W.o_chain_dp(ax, ay, **prm)
            ru, rbx, rby = W.r_chain_dp(ax, ay, **prm)
            assert np.array_equal(ou, ru) and np.array_equal(obx, rbx) and np.array_equal(oby, rby)


def test_exts2_oracle_vs_reference():
    """wmo_ksw_exts2 (oracle/wm_oracle.c) against the reference's own ksw_exts2_sse (src/ksw2_exts2_sse.c) on transcript-like inputs: every
    splice flag, left / right gap alignment, approximate and exact maximum, extension-only, reversed operands, junction annotation, N."""
    _need_ref()
    import kswcases
    n_intron = 0
    for c in kswcases.make_splice_cases(3, 400):
        kw = dict(mat=W.simple_mat(c["a"], c["b"], 1), q=c["q_"], e=c["e"], q2=c["q2"], noncan=c["noncan"], zdrop=c["zdrop"],
                  junc_bonus=c["junc_bonus"], flag=c["flag"], junc=c["junc"])
        o = W.o_ksw_exts2(c["q"], c["t"], **kw)
        r = W.r_ksw_exts2(c["q"], c["t"], **kw)
        for k in W.EZ_FIELDS:
            assert o[k] == r[k], (k, o[k], r[k], hex(c["flag"]), len(c["q"]), len(c["t"]))
        assert np.array_equal(o["cigar"], r["cigar"]), (hex(c["flag"]), W.cigar_str(o["cigar"]), W.cigar_str(r["cigar"]))
        n_intron += any((int(x) & 0xf) == 3 for x in o["cigar"])
    assert n_intron > 100, n_intron                      # the cases do exercise the intron state
