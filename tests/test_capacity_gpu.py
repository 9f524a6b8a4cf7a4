"""GPU capacity test: a 1 Gbase reference (BASELINE configs 4 / 5 are 3 Gbase: 32-bit offsets, arena sizing and the device -W counter must not be
first exercised on the 8-GPU box). The -W list is counted on the device, the index is built on the device, 256 ONT-profile reads are mapped, and
every record is compared with the reference binary on the same inputs."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from winnowmap_amd import gpu, parity, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "winnowmap_ref")


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/winnowmap_ref not built")
def test_one_gigabase_reference_device_index_and_kmer_list_parity_with_the_reference():
    tmp = tempfile.mkdtemp()
    ref = synth.make_reference(4, 250_000_000, 61, repeat_frac=0.05)          # 4 x 250 Mb: contig offsets beyond 2^29, positions beyond 2^27
    fa = os.path.join(tmp, "ref.fa")
    synth.write_fasta(fa, ref, prefix="chr")
    reads, _ = synth.make_reads(ref, 256, 12000, 62, profile="ont", sv_frac=0.05)
    # some reads from the very end of the last contig (the largest offsets into the packed reference)
    tail = ref[3][-13000:-1000].copy()
    reads[0] = synth.mutate_codes(tail, np.random.default_rng(1), 0.03, 0.0, 0.0)[:12000]
    rq = os.path.join(tmp, "reads.fa")
    with open(rq, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b">r%d\n" % i + synth.codes_to_ascii(r) + b"\n")
    seqs = [synth.codes_to_ascii(r) for r in reads]
    del ref
    ctx = gpu.Context(0, 40 << 30)
    kf = os.path.join(tmp, "rep.txt")
    n_k, st = gpu.write_repetitive_kmers_gpu(ctx, fa, 15, kf)
    assert n_k > 0
    idx, ist = gpu.Index.build_on_device(ctx, fa, kf, k=15, w=50, n_threads=16)
    assert idx.n_minimizers > 30_000_000
    idx.upload(ctx)
    m = gpu.Mapper(ctx, idx, "map-ont", gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG)
    m.set_threads(8, 8 << 30)
    ours, hits, _, _ = m.map(["r%d" % i for i in range(len(seqs))], seqs)
    p = subprocess.run([REF_BIN, "-t", "16", "-W", kf, "-cx", "map-ont", fa, rq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-1000:]
    d = parity.diff_texts(p.stdout, ours, sam=False)
    assert d["reads"] >= 250 and d["hits"] >= 250 and d["mismatches"] == 0, d
    m.close(); idx.close(); ctx.close()
