"""Shared pieces of the end-to-end parity tests: golden fixtures + regenerated inputs."""
import importlib.util
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
make_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_golden)
CASES = list(make_golden.CASES)


def golden(name):
    z = np.load(os.path.join(HERE, "golden", "e2e_%s.npz" % name))
    return z["hits"], z["cigars"], z["first"]


def compare(name, hits, cigars, first):
    """hits/cigars/first as produced by our mapper for the case's reads (MAPQ column is ignored: the reference
    does not reproduce it itself — uninitialised rep_len, src/map.c:281)."""
    gh, gc, gf = golden(name)
    assert np.array_equal(first, gf), (name, "hit counts per read differ", first.tolist(), gf.tolist())
    h = hits.copy()
    h[:, 6] = 0
    bad = np.nonzero((h != gh).any(axis=1))[0]
    assert len(bad) == 0, (name, "first differing hit", int(bad[0]), h[bad[0]].tolist(), gh[bad[0]].tolist())
    assert np.array_equal(cigars, gc), (name, "CIGAR ops differ")


def edge_case_reads(reads):
    """Reads the MCAS procedure treats specially (empty, < k, all N, N runs, unrelated, around the 10 kb MCAS gate, chimeric,
    lower case), derived from the golden 'ont' reads."""
    from winnowmap_amd import synth
    rng = np.random.default_rng(77)
    base = reads[0]
    junk = synth.codes_to_ascii(rng.integers(0, 4, 12000).astype(np.uint8))
    with_n = bytearray(reads[1])
    with_n[3000:3400] = b"N" * 400
    with_n[9000] = ord("N")
    return [b"", b"ACGT", b"ACGTACGTACGTAC", b"N" * 500, b"N" * 12000, junk[:3000], junk, base[:9999], base[:10000], base[:10001],
            base[2000:4500], bytes(with_n), base[:7000] + junk[:6000], reads[2].lower()]
