"""Differential fuzz of the HOST side of the path against the reference's library (tools/host_diff_fuzz.py, here as a test): the product's host
mapper (MCAS glue, hit.c / align.c restatement, splice mode) on oracle-backed device operations vs refshim_map (oracle/_ref), hit by hit incl. MAPQ
wherever the reference is deterministic (below the 10 kb MCAS gate, splice mode: mm_set_mapq, src/hit.c:463-508) and CIGAR by CIGAR, over random
references / reads / presets. >= 300 reads."""
import importlib.util
import os
import pytest
import wmtest as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not W.have_ref(), reason="oracle/_ref not built")
def test_host_mapper_equals_the_reference_library_on_random_configurations():
    spec = importlib.util.spec_from_file_location("host_diff_fuzz", os.path.join(ROOT, "tools", "host_diff_fuzz.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    n, bad, with_mapq = fz.run(1001, 28, verbose=True)
    assert n >= 300 and bad == 0 and with_mapq >= 150, (n, bad, with_mapq)
