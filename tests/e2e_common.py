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
