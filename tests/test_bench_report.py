"""bench.py's JSON line is assembled by a pure function: check the contract's keys for every kind of dominant kernel class."""
import argparse
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_report_has_contract_keys_for_any_dominant_class():
    B = _bench()
    args = argparse.Namespace(steps=3, warmup=1, reads_per_step=16384, read_len=15000, ref_mb=250.0)
    zero = {k: (0.0, 0.0, 0) for k in range(16)}
    for dom in (0, 5, 10, 12, 13, 14, 15):
        after = dict(zero)
        after[dom] = (500.0, 5e10, 40)
        after[4] = (100.0 if dom != 4 else 900.0, 2e10, 10)
        r = B.make_report(args, 1, 64, 10.0, 16384 * 15000 * 3.0, 1234, zero, after)
        json.dumps(r)
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
            assert key in r
        assert r["metric"] == "mapped Gbp/s" and r["unit"] == "Gbp/s" and r["vs_baseline"] is None and r["scaling"] == "weak"
        ro = r["roofline"]
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert key in ro
        assert ro["kernel"] == B.ksw_class_name(dom if dom != 4 else 4)
        assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12
        assert abs(r["value"] - 16384 * 15000 * 3.0 / 10.0 / 1e9) < 1e-12
