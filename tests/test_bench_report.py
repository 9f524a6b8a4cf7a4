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
    args = argparse.Namespace(steps=3, warmup=1, reads_per_step=16384, read_len=15000, ref_mb=250.0, config=2)
    n_classes = 52                     # WM_KSW_NCLASS (winnowmap_amd/csrc/ksw_plan.h): what Mapper.kernel_stats() returns
    zero = {k: (0.0, 0.0, 0) for k in range(n_classes)}
    assert B.ksw_class_name(0) == "ksw_dpp_kernel<4, false, false, false>" and B.ksw_class_name(14) == "ksw_dpp_kernel<8, true, false, true>"
    assert B.ksw_class_name(21) == "ksw_pmulti_kernel<4, 4>" and B.ksw_class_name(24) == "ksw_pmulti_kernel<4, 8>"
    assert B.ksw_class_name(28) == "ksw_stripe_kernel<2, 4, false, false>" and B.ksw_class_name(28 + 2 * 4 + 3) == "ksw_stripe_kernel<4, 8, true, true>"
    for dom in range(n_classes):
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


def test_gpus_n_without_gpus_fails_loudly():
    """`python bench.py --gpus 2` on a box with fewer GPUs must not silently run one rank."""
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert p.returncode != 0 and b"--gpus 2 requested" in p.stderr and b'"metric"' not in p.stdout
    # a torch.distributed environment whose size disagrees with --gpus is refused as well
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, timeout=300)
    assert p.returncode != 0 and b"WORLD_SIZE=1" in p.stderr


def test_parity_diff_masks_mapq_and_rl_only_on_the_mcas_path():
    from winnowmap_amd import parity
    # r1, r2: reads of >= 10 000 bases (the reference's MAPQ / rl:i come from an uninitialised rep_len there: masked); r4: below the gate (compared)
    ref = (b"r1\t12000\t0\t90\t+\tc\t1000\t5\t95\t80\t90\t60\ttp:A:P\trl:i:7\tcg:Z:90M\nr2\t10000\t0\t40\t-\tc\t1000\t5\t45\t30\t40\t3\tcg:Z:40M\n"
           b"r4\t9999\t0\t40\t-\tc\t1000\t5\t45\t30\t40\t33\trl:i:12\tcg:Z:40M\n")
    ours = (b"r2\t10000\t0\t40\t-\tc\t1000\t5\t45\t30\t40\t0\tcg:Z:40M\nr1\t12000\t0\t90\t+\tc\t1000\t5\t95\t80\t90\t0\ttp:A:P\trl:i:0\tcg:Z:90M\n"
            b"r4\t9999\t0\t40\t-\tc\t1000\t5\t45\t30\t40\t33\trl:i:12\tcg:Z:40M\n")
    d = parity.diff_texts(ref, ours)
    assert d["mismatches"] == 0 and d["hits"] == 3 and d["reads"] == 3 and d["mapq_compared"] == 1
    d = parity.diff_texts(ref, ours.replace(b"cg:Z:90M", b"cg:Z:89M1I"))
    assert d["mismatches"] == 1 and d["examples"][0]["read"] == "r1"
    d = parity.diff_texts(ref, ours + b"r3\t50\t0\t40\t-\tc\t1000\t5\t45\t30\t40\t0\n")
    assert d["mismatches"] == 1
    # below the gate a different MAPQ or rl:i is a mismatch; with mcas_gate=None (splice mode) it is one everywhere
    assert parity.diff_texts(ref, ours.replace(b"\t33\trl:i:12", b"\t32\trl:i:12"))["mismatches"] == 1
    assert parity.diff_texts(ref, ours.replace(b"rl:i:12", b"rl:i:11"))["mismatches"] == 1
    assert parity.diff_texts(ref, ours, mcas_gate=None)["mismatches"] == 2
    # SAM: the read length comes from the CIGAR incl. clips
    assert parity.sam_query_len(b"5S90M3I2D10H") == 108
    sam_ref = b"@SQ\tSN:c\tLN:1000\nq\t0\tc\t6\t60\t9000M1000S\t*\t0\t0\t*\t*\tNM:i:0\trl:i:5\nq2\t0\tc\t6\t17\t9000M\t*\t0\t0\t*\t*\tNM:i:0\n"
    sam_our = b"q\t0\tc\t6\t0\t9000M1000S\t*\t0\t0\t*\t*\tNM:i:0\trl:i:0\nq2\t0\tc\t6\t17\t9000M\t*\t0\t0\t*\t*\tNM:i:0\n"
    d = parity.diff_texts(sam_ref, sam_our, sam=True)
    assert d["mismatches"] == 0 and d["mapq_compared"] == 1
    assert parity.diff_texts(sam_ref, sam_our.replace(b"\t17\t", b"\t16\t"), sam=True)["mismatches"] == 1
