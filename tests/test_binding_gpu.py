"""GPU: (1) the reference's own CLI with its mapping entry point bound to libwmgpu.so (oracle/_ref/winnowmap_wm = the reference's
main + all its objects + oracle/wm_binding.cpp, linked with -Wl,--wrap=mm_map_file; INTEGRATION.md Level 0/1) must print what the
untouched reference prints; (2) bit-exact parity AT SCALE — >= 10^3 reads per BASELINE-shaped configuration — against the reference
binary on the same box (parity had only been shown on a few dozen reads)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from winnowmap_amd import gpu, parity, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "winnowmap_ref")
WM_BIN = os.path.join(ROOT, "oracle", "_ref", "winnowmap_wm")
need_ref = pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/winnowmap_ref not built")


def _write_reads(path, reads, prefix="r"):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b">%s%d\n" % (prefix.encode(), i))
            f.write(synth.codes_to_ascii(r))
            f.write(b"\n")


def _run(binary, args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([binary] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=1200)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    return p.stdout


@need_ref
@pytest.mark.skipif(not os.path.exists(WM_BIN), reason="oracle/_ref/winnowmap_wm not built")
def test_reference_cli_bound_to_the_library_prints_the_references_output():
    tmp = tempfile.mkdtemp()
    ref = synth.make_reference(2, 400000, 31, repeat_frac=0.08)
    fa = os.path.join(tmp, "ref.fa")
    synth.write_fasta(fa, ref, prefix="chr")
    km, cnt = synth.repetitive_kmers(ref, 15)
    kf = os.path.join(tmp, "rep.txt")
    synth.write_kmer_list(kf, km, cnt, 15)
    reads, _ = synth.make_reads(ref, 40, 12000, 32, profile="ont", sv_frac=0.2)
    reads += synth.make_reads(ref, 12, 3000, 33, profile="ont")[0]            # below the 10 kb MCAS gate
    rq = os.path.join(tmp, "reads.fa")
    _write_reads(rq, reads)
    # PAF+CIGAR, SAM, individual options on top of the preset (max_sw_mat has no reachable command-line switch in the reference — its option
    # table lacks --cap-sw-mat although src/main.c:235 handles it; tests/test_e2e_host.py compares it through the reference's library instead)
    for fmt, extra in (("-cx", []), ("-ax", []), ("-cx", ["-N", "3", "-p", "0.5", "--cs"])):
        args = ["-t", "4", "-W", kf] + extra + [fmt, "map-ont", fa, rq]
        sam = fmt == "-ax"
        want = _run(REF_BIN, args)
        got = _run(WM_BIN, args)
        d = parity.diff_texts(want, got, sam=sam)
        assert d["reads"] >= 40 and d["hits"] >= 40 and d["mismatches"] == 0, (fmt, extra, d)
        if sam:            # header: same @SQ lines (the @PG line carries the program path)
            hw = [l for l in want.split(b"\n") if l.startswith(b"@SQ")]
            hg = [l for l in got.split(b"\n") if l.startswith(b"@SQ")]
            assert hw == hg and len(hw) == 2
        # the same binary on its CPU path (WM_BACKEND=cpu) is the reference (MAPQ / rl:i differ from run to run in the reference itself)
        if fmt == "-cx" and not extra:
            assert parity.diff_texts(want, _run(WM_BIN, args, env={"WM_BACKEND": "cpu"}), sam=False)["mismatches"] == 0
    # splice mode through the same binding (every mm_mapopt_t field incl. noncan / junc_bonus / anchor_ext_* is handed over)
    tr = synth.make_transcripts(ref, 60, 34)
    fa2 = os.path.join(tmp, "ref2.fa")
    synth.write_fasta(fa2, ref, prefix="chr")            # (make_transcripts planted splice signals in `ref`)
    rq2 = os.path.join(tmp, "tr.fa")
    _write_reads(rq2, tr, prefix="t")
    args = ["-t", "4", "-cx", "splice", fa2, rq2]
    d = parity.diff_texts(_run(REF_BIN, args), _run(WM_BIN, args), sam=False, mcas_gate=None)      # splice mode: MAPQ and rl:i compared on every record
    assert d["reads"] >= 50 and d["mismatches"] == 0 and d["mapq_compared"] >= 50, d


def _at_scale(preset, k, w, ref, reads, kmer_list, threads=16):
    tmp = tempfile.mkdtemp()
    fa = os.path.join(tmp, "ref.fa")
    synth.write_fasta(fa, ref, prefix="ctg")
    kf = None
    if kmer_list:
        km, cnt = synth.repetitive_kmers(ref, k)
        kf = os.path.join(tmp, "rep.txt")
        synth.write_kmer_list(kf, km, cnt, k)
    rq = os.path.join(tmp, "reads.fa")
    _write_reads(rq, reads)
    want = _run(REF_BIN, ["-t", str(threads)] + (["-W", kf] if kf else []) + ["-cx", preset, fa, rq])
    ctx = gpu.Context(0, 24 << 30)
    idx = gpu.Index(fa, kf, k=k, w=w, n_threads=threads)
    idx.upload(ctx)
    m = gpu.Mapper(ctx, idx, preset, gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG)
    m.set_threads(threads, 24 << 30)
    names = [b"r%d" % i for i in range(len(reads))]
    text, hits, _, _ = m.map(names, [synth.codes_to_ascii(r) for r in reads])
    # MAPQ / rl:i take part wherever the reference assigns rep_len: below the MCAS gate, and above it for the reads the mapper reports (rescan / fallback)
    d = parity.diff_texts(want, text, sam=False, defined=parity.defined_names(names, m.rep_len_defined()))
    m.close(); idx.close(); ctx.close()
    return d, len(hits)


@need_ref
def test_parity_at_scale_config2_shape_ont():
    """BASELINE config 2 shape: 2 048 x 15 kb ONT reads (1 % with an SV) vs a 20 Mb repeat-rich reference, -W, map-ont."""
    ref = synth.make_reference(2, 10_000_000, 3, repeat_frac=0.10)
    reads, _ = synth.make_reads(ref, 2048, 15000, 4, profile="ont", sv_frac=0.01)
    d, nh = _at_scale("map-ont", 15, 50, ref, reads, True)
    assert d["reads"] == 2048 and d["hits"] >= 2048 and d["mismatches"] == 0, d
    # round 6: all 2 048 reads are above the MCAS gate; MAPQ and rl:i are compared for those whose stage-1 pass left a stretch unmapped or found nothing —
    # there the reference assigns rep_len (src/map.c:808-813, 859-861) before mm_set_mapq reads it (:933)
    assert d["mapq_compared"] >= 50, d                 # (73 of 2 062 records on this workload: 1 % of the reads carry an SV)


@need_ref
def test_parity_below_the_mcas_gate_including_mapq():
    """600 ONT reads of 1.5 .. 9.9 kb (below mm_mapopt_t::SVawareMinReadLength: the one-stage path, where the reference sets rep_len, src/map.c:859-861):
    every PAF column INCLUDING MAPQ (mm_set_mapq, src/hit.c:463-508: fp32 + logf) and the rl:i tag against the reference binary, -W list, repeats, SVs."""
    ref = synth.make_reference(2, 5_000_000, 13, repeat_frac=0.15)
    reads = []
    for i, L in enumerate((1500, 3000, 5000, 7000, 9000, 9999)):
        reads += synth.make_reads(ref, 100, L, 14 + i, profile="ont", sv_frac=0.1)[0]
    d, nh = _at_scale("map-ont", 15, 50, ref, reads, True)
    assert d["reads"] >= 590 and d["hits"] >= 600 and d["mismatches"] == 0 and d["mapq_compared"] >= 600, d      # (a read or two of 600 map nowhere, in both)


@need_ref
def test_parity_at_scale_config3_shape_hifi():
    """BASELINE config 3 shape: 1 024 x 20 kb HiFi reads vs the same kind of reference, map-pb."""
    ref = synth.make_reference(2, 10_000_000, 3, repeat_frac=0.10)
    reads, _ = synth.make_reads(ref, 1024, 20000, 5, profile="hifi")
    d, nh = _at_scale("map-pb", 15, 50, ref, reads, True)
    assert d["reads"] == 1024 and d["hits"] >= 1024 and d["mismatches"] == 0, d


@need_ref
def test_parity_at_scale_config5_shape_asm20():
    """BASELINE config 5 shape: 50 x 1 Mb contigs at 5 % divergence with SVs vs the reference they were drawn from, asm20 (k = 19; w stays 50, src/options.c:112-115)."""
    ref = synth.make_reference(3, 20_000_000, 6, repeat_frac=0.10)
    rng = np.random.default_rng(8)
    contigs = []
    for i in range(50):
        c = ref[int(rng.integers(0, 3))]
        st = int(rng.integers(0, len(c) - 1_000_000))
        q = synth.mutate_codes(c[st:st + 1_000_000].copy(), rng, 0.03, 0.01, 0.01)
        for _ in range(10):                                                   # 1 SV / 100 kb: a 2 kb deletion or a 1 kb insertion
            p = int(rng.integers(10000, len(q) - 10000))
            q = np.concatenate([q[:p], q[p + 2000:]]) if rng.integers(0, 2) else np.concatenate([q[:p], synth.random_codes(1000, rng), q[p:]])
        contigs.append(q if i % 2 else synth.revcomp_codes(q))
    d, nh = _at_scale("asm20", 19, 50, ref, contigs, False)
    assert d["reads"] == 50 and d["hits"] >= 50 and d["mismatches"] == 0, d


SUBST_BIN = os.path.join(ROOT, "oracle", "_ref", "winnowmap_subst")


@need_ref
@pytest.mark.skipif(not os.path.exists(SUBST_BIN), reason="oracle/_ref/winnowmap_subst not built")
def test_link_level_substitutes_of_the_hot_functions():
    """mm_sketch, mm_idx_get, mm_chain_dp, ksw_extd2_sse, ksw_extz2_sse and ksw_ll_qinit / ksw_ll_i16 with the reference's exact signatures, defined by oracle/wm_subst.cpp
    over the batched device operations and LINKED INTO THE REFERENCE in place of its own (oracle/_ref/winnowmap_subst; the originals are renamed
    ref_*): the reference's index build, mm_map_frag and mm_align_skeleton then run unchanged on the device kernels, one launch per call. The
    output must be the reference's. Small input: scalar calls are what the batched entry points exist to avoid."""
    tmp = tempfile.mkdtemp()
    ref = synth.make_reference(2, 200000, 41, repeat_frac=0.08)
    fa = os.path.join(tmp, "ref.fa")
    synth.write_fasta(fa, ref, prefix="chr")
    km, cnt = synth.repetitive_kmers(ref, 15)
    kf = os.path.join(tmp, "rep.txt")
    synth.write_kmer_list(kf, km, cnt, 15)
    reads = synth.make_reads(ref, 3, 11000, 42, profile="ont", sv_frac=0.3)[0] + synth.make_reads(ref, 5, 3000, 43, profile="ont")[0]
    rng = np.random.default_rng(44)
    for g in range(2):                                                       # an inversion in the middle: the z-drop test's ksw_ll_i16 (src/align.c:72-86) and mm_align1_inv
        src = ref[g][50000 + 20000 * g:50000 + 20000 * g + 12000].copy()
        src[5000:5600] = synth.revcomp_codes(src[5000:5600])
        reads.append(synth.mutate_codes(src, rng, 0.02, 0.01, 0.01))
    rq = os.path.join(tmp, "reads.fa")
    _write_reads(rq, reads)
    args = ["-t", "2", "-W", kf, "-cx", "map-ont", fa, rq]
    want = _run(REF_BIN, args)
    got = _run(SUBST_BIN, args)
    d = parity.diff_texts(want, got, sam=False)
    assert d["reads"] == 10 and d["hits"] >= 10 and d["mismatches"] == 0, d
    assert parity.diff_texts(want, _run(SUBST_BIN, args, env={"WM_SUBST": "off"}), sam=False)["mismatches"] == 0


@need_ref
@pytest.mark.skipif(not os.path.exists(SUBST_BIN), reason="oracle/_ref/winnowmap_subst not built")
def test_splice_mode_through_the_substituted_ksw_exts2():
    """`-ax splice` of the reference with ksw_exts2_sse (src/ksw2.h:63-64, called at src/align.c:326-327) replaced by the one-job form of
    wm_ksw_exts2_batch (oracle/wm_subst.cpp): spliced reads with canonical and non-canonical introns, both strands; the records — CIGARs with
    N operations, ts:A tags — must be the reference's. (Sketching runs on the device as well; cDNA chaining stays with the reference.)"""
    tmp = tempfile.mkdtemp()
    ref = synth.make_reference(2, 300000, 51, repeat_frac=0.0)
    reads = synth.make_transcripts(ref, 60, 7)
    fa = os.path.join(tmp, "ref.fa")
    synth.write_fasta(fa, ref, prefix="chr")
    rq = os.path.join(tmp, "reads.fa")
    _write_reads(rq, reads, prefix="t")
    args = ["-t", "2", "-cx", "splice", fa, rq]
    want = _run(REF_BIN, args)
    got = _run(SUBST_BIN, args)
    d = parity.diff_texts(want, got, sam=False, mcas_gate=None)
    assert d["reads"] == 60 and d["hits"] >= 55 and d["mismatches"] == 0, d
    assert sum(1 for ln in got.decode().splitlines() if "N" in ln.split("cg:Z:")[-1].split("\t")[0]) >= 50


@need_ref
def test_splice_mode_of_the_mapper_matches_the_reference_cli():
    """`-cx splice` (src/options.c:116-128) through wm_map_reads: cDNA chaining cost in the window kernels, every alignment through
    wm_ksw_exts2_batch, two passes per region (one per transcript strand, src/align.c:884-904), ts:A and N in the records. Compared with the
    untouched reference binary, then once more with -uf-like options (one pass) through the option struct."""
    tmp = tempfile.mkdtemp()
    ref = synth.make_reference(2, 300000, 41, repeat_frac=0.05)
    reads = synth.make_transcripts(ref, 200, 42)
    fa = os.path.join(tmp, "ref.fa")
    synth.write_fasta(fa, ref, prefix="chr")
    rq = os.path.join(tmp, "reads.fa")
    _write_reads(rq, reads)
    ctx = gpu.Context(0, 8 << 30)
    idx = gpu.Index(fa, None, k=15, w=25, n_threads=8)
    idx.upload(ctx)
    names = [b"r%d" % i for i in range(len(reads))]
    seqs = [synth.codes_to_ascii(r) for r in reads]
    for preset in ("splice", "splice:hq"):
        want = _run(REF_BIN, ["-t", "8", "-k", "15", "-w", "25", "-cx", preset, fa, rq])
        m = gpu.Mapper(ctx, idx, preset, gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG)
        m.set_threads(8, 4 << 30)
        text, hits, _, _ = m.map(names, seqs)
        d = parity.diff_texts(want, text, sam=False, mcas_gate=None)              # MAPQ (mm_set_mapq, src/hit.c:463-508) and rl:i included
        assert d["reads"] >= 180 and d["hits"] >= 180 and d["mismatches"] == 0 and d["mapq_compared"] >= 180, (preset, d)
        import re
        assert text.count(b"ts:A:+") > 20 and text.count(b"ts:A:-") > 20 and len(re.findall(rb"[0-9]N", text)) > 200
        m.close()
    idx.close(); ctx.close()


@need_ref
def test_splice_mode_with_a_junction_annotation_matches_the_reference_library():
    """--junc-bed (src/main.c:416) through wm_index_read_junc_bed + wm_map_reads: the junction bits ride with every exts2 job to the device
    (wm_ksw_exts2_batch's `junc`). The reference's CLI cannot be the yardstick — its BED reader crashes (oracle/ref_shim.cpp) — so the same
    intervals are injected into the reference's index and its library maps the reads (refshim_map), hit by hit and CIGAR by CIGAR."""
    import ctypes as C
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import wmtest as W
    from test_e2e_host import _transcripts_with_bed, _bed_introns
    R = W.ref()
    tmp = tempfile.mkdtemp()
    ref = synth.make_reference(2, 300000, 51, repeat_frac=0.05)
    bed = os.path.join(tmp, "anno.bed")
    reads = _transcripts_with_bed(ref, 80, 52, bed)
    fa = os.path.join(tmp, "ref.fa")
    synth.write_fasta(fa, ref, prefix="ref")
    seqs = [synth.codes_to_ascii(r) for r in reads]
    mi = R.refshim_idx_build(fa.encode(), b"", 15, 25, 4)
    R.refshim_idx_set_junc.argtypes = [C.c_void_p, C.c_int, C.c_int, W.i32p, W.i32p, W.i32p]
    for ctg, iv in _bed_introns(bed, ["ref0", "ref1"]).items():
        a = np.array(iv, np.int32).reshape(-1, 3)
        assert R.refshim_idx_set_junc(mi, ctg, len(a), np.ascontiguousarray(a[:, 0]), np.ascontiguousarray(a[:, 1]), np.ascontiguousarray(a[:, 2])) == 0
    opt = R.refshim_mapopt(b"splice", 0x4 | 0x20, mi)
    ctx = gpu.Context(0, 8 << 30)
    idx = gpu.Index(fa, None, k=15, w=25, n_threads=8)
    idx.read_junc_bed(bed)
    idx.upload(ctx)
    m = gpu.Mapper(ctx, idx, "splice", gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG)
    m.set_threads(8, 4 << 30)
    text, hits, cigs, first = m.map([b"r%d" % i for i in range(len(seqs))], seqs)
    n_hits = n_introns = 0
    for i, s in enumerate(seqs):
        rh = np.zeros(16 * 256, np.int32); rc = np.zeros(400000, np.uint32); rnc = C.c_int64()
        rn = R.refshim_map(mi, opt, s, len(s), b"q", rh, 256, rc, len(rc), C.byref(rnc))
        ours = hits[first[i]:first[i + 1]].copy(); want = rh[:16 * rn].reshape(-1, 16).copy()
        assert np.array_equal(ours, want), (i, ours.tolist(), want.tolist())
        c0 = int(hits[:first[i], 7].sum()); c1 = c0 + int(ours[:, 7].sum())
        assert np.array_equal(cigs[c0:c1], rc[:rnc.value]), i
        n_hits += rn
        n_introns += int(np.sum((rc[:rnc.value] & 0xf) == 3))
    assert n_hits >= 70 and n_introns >= 100, (n_hits, n_introns)
    m.close(); idx.close(); ctx.close()


@need_ref
def test_homopolymer_compressed_index_maps_like_the_reference_with_H():
    """`-H` (MM_I_HPC; src/sketch.c:152-163, mm_adjust_minier's HPC branch src/align.c:352-361): minimizers over the homopolymer-compressed sequence on both
    sides. A reference with long homopolymer runs planted, reads with run-length errors on top of the ONT profile, long reads (MCAS) and short ones (MAPQ
    compared); the index built on the host and on the device must be the same index; PAF + CIGAR against `winnowmap_ref -H`, then the reference's own CLI
    bound to the library (`winnowmap_wm -H`: the index travels as an .mmi file with the flag in its header)."""
    tmp = tempfile.mkdtemp()
    rng = np.random.default_rng(91)
    ref = synth.make_reference(2, 400000, 90, repeat_frac=0.05)
    for c in ref:                                          # homopolymer runs of 5 .. 300 bases every few kb
        for p in range(1500, len(c) - 2000, 3000):
            n = int(rng.choice([5, 9, 20, 60, 300]))
            c[p:p + n] = int(rng.integers(0, 4))
    fa = os.path.join(tmp, "ref.fa")
    synth.write_fasta(fa, ref, prefix="chr")
    km, cnt = synth.repetitive_kmers(ref, 15)
    kf = os.path.join(tmp, "rep.txt")
    synth.write_kmer_list(kf, km, cnt, 15)
    reads, _ = synth.make_reads(ref, 60, 12000, 92, profile="ont", sv_frac=0.1)
    reads += synth.make_reads(ref, 60, 3000, 93, profile="ont")[0]

    def hp_errors(r):                                      # lengthen / shorten runs, the error HPC was invented for
        out = []
        i = 0
        while i < len(r):
            j = i + 1
            while j < len(r) and r[j] == r[i]:
                j += 1
            n = j - i
            if n >= 3 and rng.random() < 0.3:
                n = max(1, n + int(rng.integers(-2, 3)))
            out.append(np.full(n, r[i], np.uint8))
            i = j
        return np.concatenate(out)

    reads = [hp_errors(r) for r in reads]
    rq = os.path.join(tmp, "reads.fa")
    _write_reads(rq, reads)
    args = ["-t", "4", "-H", "-W", kf, "-cx", "map-ont", fa, rq]
    want = _run(REF_BIN, args)
    ctx = gpu.Context(0, 8 << 30)
    ih = gpu.Index(fa, kf, 15, 50, hpc=True)
    idv, _ = gpu.Index.build_on_device(ctx, fa, kf, 15, 50, hpc=True)
    a, b = os.path.join(tmp, "h.mmi"), os.path.join(tmp, "d.mmi")
    ih.save(a); idv.save(b)
    assert open(a, "rb").read() == open(b, "rb").read() and ih.n_minimizers > 1000
    i0 = gpu.Index(fa, kf, 15, 50)
    assert ih.n_minimizers != i0.n_minimizers                     # (it is another index than the plain one)
    i0.close()
    idv.upload(ctx)
    m = gpu.Mapper(ctx, idv, "map-ont", gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG)
    m.set_threads(8, 2 << 30)
    seqs = [synth.codes_to_ascii(r) for r in reads]
    ours, hits, _, _ = m.map([b"r%d" % i for i in range(len(seqs))], seqs)
    d = parity.diff_texts(want, ours, sam=False)
    assert d["reads"] >= 100 and d["hits"] >= 100 and d["mismatches"] == 0 and d["mapq_compared"] >= 50, d
    m.close(); ih.close(); idv.close(); ctx.close()
    if os.path.exists(WM_BIN):
        d = parity.diff_texts(want, _run(WM_BIN, args), sam=False)
        assert d["reads"] >= 100 and d["mismatches"] == 0, d


@need_ref
def test_reference_indexed_in_parts_matches_split_prefix_of_the_reference_cli():
    """wm_index_build_parts + wm_map_file_split against `winnowmap_ref -t 1 -I 350k --split-prefix …` (mm_split_merge, src/map.c:1050-1105):
    six 200-kb contigs = three index parts, a 30-kb duplication across two parts, 120 reads; the merged records must be the reference's."""
    tmp = tempfile.mkdtemp()
    ref = synth.make_reference(6, 200000, 81, repeat_frac=0.08)
    ref[5][20000:50000] = ref[0][60000:90000]
    fa = os.path.join(tmp, "ref.fa")
    synth.write_fasta(fa, ref, prefix="chr")
    km, cnt = synth.repetitive_kmers(ref, 15)
    kf = os.path.join(tmp, "rep.txt")
    synth.write_kmer_list(kf, km, cnt, 15)
    reads, _ = synth.make_reads(ref, 90, 12000, 82, profile="ont", sv_frac=0.1)
    reads += synth.make_reads(ref, 26, 3000, 83, profile="ont")[0]
    rng = np.random.default_rng(84)
    for s0 in (61000, 66000):
        reads.append(synth.mutate_codes(ref[0][s0:s0 + 12000].copy(), rng, 0.03, 0.02, 0.02))
        reads.append(synth.revcomp_codes(synth.mutate_codes(ref[5][s0 - 40000:s0 - 28000].copy(), rng, 0.03, 0.02, 0.02)))
    rq = os.path.join(tmp, "reads.fa")
    _write_reads(rq, reads)
    want = _run(REF_BIN, ["-t", "1", "-I", "350k", "--split-prefix", os.path.join(tmp, "sp"), "-W", kf, "-cx", "map-ont", fa, rq])
    ctx = gpu.Context(0, 8 << 30)
    parts = gpu.build_index_parts(fa, kf, 15, 50, 350000)
    assert len(parts) == 3
    opt, _, _ = gpu.mapopt_preset("map-ont")
    opt.flag |= gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG
    outp = os.path.join(tmp, "ours.paf")
    st = gpu.map_file_split(ctx, parts, opt, 8, rq, outp)
    assert st["reads"] == len(reads)
    d = parity.diff_texts(want, open(outp, "rb").read(), sam=False)
    assert d["reads"] >= 100 and d["hits"] >= 110 and d["mismatches"] == 0 and d["mapq_compared"] >= 26, d
    ours = open(outp, "rb").read()
    # one part in memory at a time (src/main.c:398-429): the parts built when their turn comes — on the host, then on the device — and a run fed part by part
    for on_dev in (False, True):
        out2 = os.path.join(tmp, "lazy%d.paf" % on_dev)
        st2 = gpu.map_file_split_fasta(ctx, fa, kf, 15, 50, 350000, opt, 8, rq, out2, on_device=on_dev)
        assert st2["parts"] == 3 and st2["reads"] == len(reads)
        assert open(out2, "rb").read() == ours, "on_device=%s" % on_dev
    run = gpu.SplitRun(ctx, opt, 15, 50, 8, rq)
    for p in parts:
        run.add_part(p)
        p.close()                       # (a part is done with when add_part returns)
    out3 = os.path.join(tmp, "fed.paf")
    run.finish(out3)
    assert open(out3, "rb").read() == ours
    with pytest.raises(gpu.WmError):    # a part built with another k is refused
        r2 = gpu.SplitRun(ctx, opt, 15, 50, 8, rq)
        try:
            r2.add_part(gpu.Index(fa, kf, 17, 50))
        finally:
            r2.abort()
    ctx.close()
    # the same through the reference's own CLI bound to the library: main presents the parts, the wrapped mm_split_merge runs wm_map_file_split
    # (oracle/wm_binding.cpp); PAF and SAM (header: @PG from main, then the @SQ lines of every part)
    if os.path.exists(WM_BIN):
        for fmt in ("-cx", "-ax"):
            args = ["-t", "1", "-I", "350k", "--split-prefix", os.path.join(tmp, "sp2"), "-W", kf, fmt, "map-ont", fa, rq]
            w2, g2 = _run(REF_BIN, args), _run(WM_BIN, args)
            d = parity.diff_texts(w2, g2, sam=fmt == "-ax")
            assert d["reads"] >= 100 and d["hits"] >= 110 and d["mismatches"] == 0, (fmt, d)
            if fmt == "-ax":
                hw = [l.split(b"\t")[0] for l in w2.split(b"\n") if l.startswith(b"@")]
                hg = [l.split(b"\t")[0] for l in g2.split(b"\n") if l.startswith(b"@")]
                assert hw == hg and hw[0] == b"@PG" and hw.count(b"@SQ") == 6, (hw, hg)
                assert [l for l in w2.split(b"\n") if l.startswith(b"@SQ")] == [l for l in g2.split(b"\n") if l.startswith(b"@SQ")]
