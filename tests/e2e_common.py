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


MCAS_GATE = 10000     # mm_mapopt_t::SVawareMinReadLength: reads at least this long go through the two-stage MCAS procedure (src/map.c:334)


def mask_mapq(read_len, *hit_arrays, splice=False):
    """MAPQ (column 6 of a hit row) is compared wherever the reference is deterministic: below the MCAS gate and in splice mode. On the MCAS path
    mm_set_mapq is fed an uninitialised rep_len (src/map.c:281, :933), so the reference does not reproduce its own MAPQ there: zeroed on both sides."""
    if not splice and read_len >= MCAS_GATE:
        for h in hit_arrays:
            h[:, 6] = 0


def compare(name, hits, cigars, first, reads=None):
    """hits/cigars/first as produced by our mapper for the case's reads. The fixture holds MAPQ for reads below the MCAS gate and 0 above it
    (tests/golden/make_golden.py); `reads` (their sequences) says which of our rows to zero likewise."""
    gh, gc, gf = golden(name)
    assert np.array_equal(first, gf), (name, "hit counts per read differ", first.tolist(), gf.tolist())
    h = hits.copy()
    if reads is None:
        reads = make_golden.inputs(name, __import__("tempfile").mkdtemp())[4]
    for i, s in enumerate(reads):
        mask_mapq(len(s), h[int(first[i]):int(first[i + 1])])
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
