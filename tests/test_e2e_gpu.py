"""GPU: the full product path (sketch → seed → chain → align kernels behind the C-ABI, host MCAS glue) must
reproduce the reference's hits and CIGARs bit-for-bit on the golden cases; plus PAF/SAM text sanity."""
import ctypes as C
import tempfile
import numpy as np
import pytest
import e2e_common as E
from winnowmap_amd import gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = gpu.Context(0, 8 << 30)
    yield c
    c.close()


@pytest.mark.parametrize("name", E.CASES)
def test_mapper_matches_reference_golden(ctx, name):
    tmp = tempfile.mkdtemp()
    preset, fa, kf, k, reads = E.make_golden.inputs(name, tmp)
    idx = gpu.Index(fa, kf, k=k, w=50, n_threads=8, hpc=bool(E.make_golden.IDX_FLAG.get(name, 0) & 1))      # (ont_hpc: the CLI's -H)
    idx.upload(ctx)
    m = gpu.Mapper(ctx, idx, preset, gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG)
    text, hits, cigars, first = m.map(["read%d" % i for i in range(len(reads))], reads)
    E.compare(name, hits, cigars, first, reads)
    # PAF text: one line per hit, cg:Z: equals the CIGAR ops
    lines = text.decode().strip().split("\n")
    assert len(lines) == int(first[-1])
    o = 0
    for ln, h in zip(lines, hits):
        f = ln.split("\t")
        assert int(f[2]) == h[3] and int(f[3]) == h[4] and int(f[7]) == h[1] and int(f[8]) == h[2] and f[4] == "+-"[h[5]]
        cg = [x for x in f if x.startswith("cg:Z:")][0][5:]
        ops = cigars[o:o + h[7]]
        o += h[7]
        assert cg == "".join("%d%s" % (int(c) >> 4, "MIDN"[int(c) & 0xf]) for c in ops)
    st = m.stats()
    assert st["ksw_jobs"] > 0 and st["dp_cells"] > 0
    m.close()
    idx.close()


def test_batching_is_order_independent(ctx):
    # mapping reads one by one or all together gives identical records (no cross-read state)
    tmp = tempfile.mkdtemp()
    preset, fa, kf, k, reads = E.make_golden.inputs("ont_short", tmp)
    idx = gpu.Index(fa, kf, k=k, w=50)
    idx.upload(ctx)
    m = gpu.Mapper(ctx, idx, preset, gpu.MM_F_CIGAR | gpu.MM_F_OUT_SAM)
    names = ["read%d" % i for i in range(len(reads))]
    all_text, _, _, _ = m.map(names, reads)
    one = b"".join(m.map([names[i]], [reads[i]])[0] for i in range(len(reads)))
    assert all_text == one and all_text.count(b"\n") >= len(reads)
    m.close()
    idx.close()


def test_edge_case_reads_match_reference_library(ctx):
    """Empty / tiny / all-N / unrelated / chimeric reads and reads around the 10 kb MCAS gate, in one batch with several host
    threads, against the prebuilt reference library (oracle/_ref; skipped when it was not built)."""
    import ctypes as C
    import wmtest as W
    if not W.have_ref():
        pytest.skip("oracle/_ref not built")
    R = W.ref()
    tmp = tempfile.mkdtemp()
    preset, fa, kf, k, reads = E.make_golden.inputs("ont", tmp)
    cases = E.edge_case_reads(reads)
    idx = gpu.Index(fa, kf, k=k, w=50, n_threads=8)
    idx.upload(ctx)
    m = gpu.Mapper(ctx, idx, preset, gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG)
    m.set_threads(8, 2 << 30)
    text, hits, cigars, first = m.map(["q%d" % i for i in range(len(cases))], cases)
    mi = R.refshim_idx_build(fa.encode(), (kf or "").encode(), k, 50, 4)
    opt = R.refshim_mapopt(preset.encode(), 0x4 | 0x20, mi)
    co = 0
    for ci, s in enumerate(cases):
        rh = np.zeros(16 * 256, np.int32); rc = np.zeros(2000000, np.uint32); rnc = C.c_int64()
        rn = R.refshim_map(mi, opt, s, len(s), b"q", rh, 256, rc, len(rc), C.byref(rnc))
        a = hits[int(first[ci]):int(first[ci + 1])].copy(); b = rh[:16 * rn].reshape(-1, 16).copy()
        assert len(a) == rn, (ci, len(s), len(a), rn)
        E.mask_mapq(len(s), a, b)
        assert np.array_equal(a, b), (ci, len(s))
        nc = int(a[:, 7].sum()) if len(a) else 0
        assert nc == rnc.value and np.array_equal(cigars[co:co + nc], rc[:nc]), (ci, len(s))
        co += nc
    m.close()
    idx.close()


def test_map_file_pipeline(ctx):
    """wm_map_file: FASTA -> PAF through overlapped reader / mapper / writer; equals mapping the same mini-batches directly."""
    import os
    tmp = tempfile.mkdtemp()
    preset, fa, kf, k, reads = E.make_golden.inputs("ont_short", tmp)
    reads = [r[:4000 + 137 * i] for i, r in enumerate(reads)]             # different lengths: the per-batch order matters
    rq = os.path.join(tmp, "reads.fa")
    with open(rq, "wb") as f:
        for i, s in enumerate(reads):
            f.write(b">read%d\n" % i + s + b"\n")
    idx = gpu.Index(fa, kf, k=k, w=50)
    idx.upload(ctx)
    m = gpu.Mapper(ctx, idx, preset, gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG)
    m.set_threads(4, 2 << 30)
    K = 20000
    outp = os.path.join(tmp, "out.paf")
    st = m.map_file(rq, outp, K)
    assert st["reads"] == len(reads) and st["batches"] >= 3
    expect = b""
    i = 0
    while i < len(reads):                                                 # the reader's mini-batches, each longest read first
        j, bases = i, 0
        while j < len(reads):
            bases += len(reads[j]); j += 1
            if bases >= K:
                break
        order = sorted(range(i, j), key=lambda t: (len(reads[t]), t), reverse=True)
        expect += m.map(["read%d" % t for t in order], [reads[t] for t in order])[0]
        i = j
    assert open(outp, "rb").read() == expect and expect.count(b"\n") >= len(reads)
    m.close()
    idx.close()


def test_workload_scale_properties(ctx):
    """Size-independent properties on a bench-shaped batch (2048 x 15 kb ONT reads vs 20 Mb with repeats, -W, several host
    threads and groups): every CIGAR consumes exactly its query and reference intervals, intervals are inside the sequences,
    the primary hit of (almost) every read lands on its true origin and strand, and mapping the batch twice is deterministic."""
    from winnowmap_amd import synth
    import os
    tmp = tempfile.mkdtemp()
    ref = synth.make_reference(2, 10_000_000, 3, repeat_frac=0.10)
    fa = os.path.join(tmp, "ref.fa")
    synth.write_fasta(fa, ref, prefix="ctg")
    kf = os.path.join(tmp, "rep.txt")
    assert gpu.write_repetitive_kmers(fa, 15, kf) > 0
    idx = gpu.Index(fa, kf, k=15, w=50, n_threads=16)
    idx.upload(ctx)
    m = gpu.Mapper(ctx, idx, "map-ont", gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG)
    m.set_threads(16, 6 << 30)
    reads, truth = synth.make_reads(ref, 2048, 15000, 4, profile="ont", sv_frac=0.01)
    seqs = [synth.codes_to_ascii(r) for r in reads]
    names = ["r%d" % i for i in range(len(seqs))]
    text, hits, cigars, first = m.map(names, seqs)
    text2, hits2, cigars2, first2 = m.map(names, seqs)
    assert text == text2 and np.array_equal(hits, hits2) and np.array_equal(cigars, cigars2)
    # CIGAR consistency
    ops = cigars & 0xf
    lens = (cigars >> 4).astype(np.int64)
    off = np.concatenate([[0], np.cumsum(hits[:, 7])])
    cq = np.concatenate([[0], np.cumsum(np.where(np.isin(ops, (0, 1, 7, 8)), lens, 0))])
    cr = np.concatenate([[0], np.cumsum(np.where(np.isin(ops, (0, 2, 3, 7, 8)), lens, 0))])
    qc, rc = cq[off[1:]] - cq[off[:-1]], cr[off[1:]] - cr[off[:-1]]
    has = hits[:, 7] > 0
    assert np.array_equal(qc[has], (hits[:, 4] - hits[:, 3])[has]) and np.array_equal(rc[has], (hits[:, 2] - hits[:, 1])[has])
    assert (hits[:, 3] >= 0).all() and (hits[:, 4] <= 15000).all() and (hits[:, 1] >= 0).all() and (hits[:, 2] <= 10_000_000).all()
    # recall against the generator's truth: the first (primary) hit of a read overlaps its origin on the right strand
    ok = 0
    for i, (ci, st, strand) in enumerate(truth):
        if first[i + 1] == first[i]:
            continue
        h = hits[first[i]]
        ok += int(h[0] == ci and h[5] == strand and h[1] < st + 17000 and h[2] > st)
    assert ok >= 0.97 * len(truth), ok
    m.close()
    idx.close()


def test_small_arena_splits_batches_and_gives_the_same_records():
    """Batches are sized by demand (the hub takes whole queues), not by HBM: with an arena far smaller than a batch needs, the sketch / seed /
    chain calls are served in halves and the ksw call in traceback-sized chunks (each uploading only its own operands). Same records as
    with a comfortable arena."""
    from winnowmap_amd import synth
    tmp = tempfile.mkdtemp()
    ref = synth.make_reference(2, 1_000_000, 51, repeat_frac=0.10)
    fa = tmp + "/ref.fa"
    synth.write_fasta(fa, ref)
    km, cnt = synth.repetitive_kmers(ref, 15)
    kf = tmp + "/rep.txt"
    synth.write_kmer_list(kf, km, cnt, 15)
    reads, _ = synth.make_reads(ref, 200, 12000, 52, profile="ont", sv_frac=0.05)
    names = [b"r%d" % i for i in range(len(reads))]
    seqs = [synth.codes_to_ascii(r) for r in reads]
    out = []
    for arena in (2 << 30, 48 << 20):
        c = gpu.Context(0, arena)
        idx = gpu.Index(fa, kf, k=15, w=50, n_threads=4)
        idx.upload(c)
        m = gpu.Mapper(c, idx, "map-ont", gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG)
        m.set_threads(4, arena)
        text, hits, cigars, first = m.map(names, seqs)
        out.append((text, hits.copy(), cigars.copy(), first.copy()))
        m.close(); idx.close(); c.close()
    assert len(out[0][1]) >= 200
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][3], out[1][3])


def test_index_handed_on_device_to_device_and_the_file_loop_over_two_mappers(ctx):
    """SURVEY §8(b) / §8(e) from C: the index goes from one device context to another without touching the host (wm_index_upload_peer: one
    hipMemcpyPeer per flat array — two contexts of this one GPU here, two GPUs on a node), a mapper on each, and wm_map_file_multi fans the
    mini-batches of a reads file over them (2 lanes per mapper): the file must equal wm_map_file's of one mapper. Then the RCCL form: a 1-rank
    `nccl` group broadcasts the flat arrays and the context receives them from the broadcast buffers (wm_index_upload_dev, winnowmap_amd/dist.py)."""
    import os
    tmp = tempfile.mkdtemp()
    preset, fa, kf, k, reads = E.make_golden.inputs("ont_short", tmp)
    reads = [r[:3000 + 211 * i] for i, r in enumerate(reads)] * 3
    rq = os.path.join(tmp, "reads.fa")
    with open(rq, "wb") as f:
        for i, s in enumerate(reads):
            f.write(b">read%d\n" % i + s + b"\n")
    idx = gpu.Index(fa, kf, k=k, w=50)
    idx.upload(ctx)
    n_dev = int(gpu.lib().wm_device_count())
    ctx2 = gpu.Context(1 if n_dev > 1 else 0, 2 << 30)          # the second GPU of the node when there is one (hipMemcpyPeer across devices), else a second context of this one
    gpu.lib().wm_ctx_device.argtypes = [C.c_void_p]
    assert gpu.lib().wm_ctx_device(ctx2._h) == (1 if n_dev > 1 else 0)          # (on a multi-GPU box this test must really cross devices — hipMemcpyPeer over xGMI)
    idx.upload_peer(ctx2, ctx)
    with pytest.raises(gpu.WmError):                            # source == destination used to free the arrays it then copied from
        idx.upload_peer(ctx, ctx)
    m1 = gpu.Mapper(ctx, idx, preset, gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG); m1.set_threads(4, 1 << 30)
    m2 = gpu.Mapper(ctx2, idx, preset, gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG); m2.set_threads(4, 1 << 30)
    one, two = os.path.join(tmp, "one.paf"), os.path.join(tmp, "two.paf")
    st1 = m1.map_file(rq, one, 15000)
    st2 = gpu.map_file_multi([m1, m2], rq, two, 15000)
    assert st1["reads"] == st2["reads"] == len(reads) and st2["batches"] >= 6
    a, b = open(one, "rb").read(), open(two, "rb").read()
    assert a == b and a.count(b"\n") >= len(reads)
    # the second context alone maps like the first (its index arrived device to device)
    t1 = m1.map(["q%d" % i for i in range(8)], reads[:8])[0]
    t2 = m2.map(["q%d" % i for i in range(8)], reads[:8])[0]
    assert t1 == t2 and len(t1) > 0
    m1.close(); m2.close()
    # RCCL: one-rank group, the broadcast buffers feed the context directly. In a process of its own with torch imported FIRST: torch brings its own HIP
    # runtime, and a process that has already initialised the system's through libwmgpu.so leaves torch without a device (bench.py imports torch first too)
    import subprocess
    import sys
    np.save(os.path.join(tmp, "t1.npy"), np.frombuffer(t1, np.uint8))
    script = """
import os, sys, tempfile
import torch, torch.distributed as dist
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import e2e_common as E
from winnowmap_amd import gpu, dist as wmdist
tmp = %r
preset, fa, kf, k, reads = E.make_golden.inputs("ont_short", tempfile.mkdtemp())
reads = [r[:3000 + 211 * i] for i, r in enumerate(reads)]
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(wmdist.free_port())
dist.init_process_group("nccl", rank=0, world_size=1)
idx = gpu.Index(fa, kf, k=k, w=50)
ctx3 = gpu.Context(0, 2 << 30)
idx_b, on_dev = wmdist.broadcast_index(idx, 0, dist, torch.device("cuda", 0), ctx3)
assert on_dev
m3 = gpu.Mapper(ctx3, idx_b, preset, gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG)
t3 = m3.map(["q%%d" %% i for i in range(8)], reads[:8])[0]
assert np.array_equal(np.frombuffer(t3, np.uint8), np.load(os.path.join(tmp, "t1.npy"))), "records differ"
m3.close(); ctx3.close(); dist.destroy_process_group()
print("RCCL-received index maps identically")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), tmp)
    p = subprocess.run([sys.executable, "-c", script], capture_output=True, timeout=300)
    assert p.returncode == 0 and b"maps identically" in p.stdout, p.stderr.decode(errors="replace")[-1500:]
    ctx2.close(); idx.close()
