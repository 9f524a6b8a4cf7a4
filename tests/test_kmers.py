"""The -W list writer of the library (wm_write_repetitive_kmers, host code of libwmgpu.so — no GPU needed) against the numpy
restatement of meryl's `print greater-than distinct=0.9998` (synth.repetitive_kmers; ext/meryl/src/meryl/merylOp-nextMer.C:103-115)."""
import os
import tempfile
import numpy as np
import pytest
from winnowmap_amd import gpu, synth


def _read_list(path, k):
    km, cnt = [], []
    for line in open(path):
        s, c = line.split()
        assert len(s) == k
        x = 0
        for ch in s:
            x = x << 2 | "ACGT".index(ch)
        km.append(x)
        cnt.append(int(c))
    o = np.argsort(np.array(km, np.uint64), kind="stable")
    return np.array(km, np.uint64)[o], np.array(cnt, np.int64)[o]


@pytest.mark.parametrize("k,distinct", [(15, 0.9998), (15, 0.99), (19, 0.9998), (11, 0.5)])
def test_repetitive_kmer_list_matches_numpy_restatement(k, distinct):
    tmp = tempfile.mkdtemp()
    ref = synth.make_reference(2, 120000, 21 + k, repeat_frac=0.15)
    fa = os.path.join(tmp, "ref.fa")
    synth.write_fasta(fa, ref)
    out = os.path.join(tmp, "rep.txt")
    n = gpu.write_repetitive_kmers(fa, k, out, distinct)
    km, cnt = synth.repetitive_kmers(ref, k, distinct)
    a, c = _read_list(out, k)
    assert n == len(km) and n > 0
    assert np.array_equal(a, km) and np.array_equal(c, cnt)


def test_threshold_follows_meryl_integer_target():
    """Truncated target and 'only counts that occur': with 3 distinct k-mers (counts 1, 1, 5) and distinct = 0.5 the target is
    int(1.5) = 1, reached by the first histogram entry (count 1, cumulative 2), so only the count-5 k-mer is printed; a target
    below 1 (tiny inputs) still takes the smallest count that occurs as the threshold."""
    codes = [np.array([0, 0, 0, 0, 0, 0, 0, 1, 2], np.uint8)]       # AAA x5 (canonical AAA), AAC, ACG
    km, cnt = synth.repetitive_kmers(codes, 3, 0.5)
    assert [synth.kmer_to_str(x, 3) for x in km] == ["AAA"] and cnt.tolist() == [5]
    tmp = tempfile.mkdtemp()
    fa = os.path.join(tmp, "t.fa")
    synth.write_fasta(fa, codes)
    out = os.path.join(tmp, "o.txt")
    assert gpu.write_repetitive_kmers(fa, 3, out, 0.5) == 1
    assert open(out).read() == "AAA\t5\n"
    assert gpu.write_repetitive_kmers(fa, 3, out, 0.1) == 1            # target int(0.3) = 0 -> threshold = smallest count present (1)
