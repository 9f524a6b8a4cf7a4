#!/usr/bin/env python
"""One-process sweep of (host threads, scheduler groups) on the bench workload: reads/s per configuration.
  python tools/tune_groups.py "32:4,64:8,64:16" [--config 2] [--reads 65536]"""
import argparse
import importlib.util
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)
from winnowmap_amd import gpu, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs")
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--ref-mb", type=float, default=250)
    ap.add_argument("--hbm-gb", type=float, default=220)
    a = ap.parse_args()
    cfg = B.CONFIGS[a.config]
    n = a.reads or cfg["reads_per_step"]
    tmp = tempfile.mkdtemp(prefix="wmtune_")
    ref, fa, kf = B.make_workload(a.ref_mb, tmp)
    idx = gpu.Index(fa, kf, k=15, w=50, n_threads=16)
    reads, _ = synth.make_reads(ref, 2 * n, cfg["read_len"], cfg["seed"], profile=cfg["profile"], sv_frac=cfg["sv_frac"])
    seqs = [synth.codes_to_ascii(r) for r in reads]
    names = [b"r%d" % i for i in range(len(seqs))]
    batches = [(names[:n], seqs[:n]), (names[n:], seqs[n:])]
    rows = []
    for spec_ in a.configs.split(","):
        t, g = (int(x) for x in spec_.split(":"))
        os.environ["WM_GROUPS"] = str(g)
        arena = int(min(48.0, a.hbm_gb / g) * (1 << 30))
        ctx = gpu.Context(0, arena)
        idx.upload(ctx)
        m = gpu.Mapper(ctx, idx, cfg["preset"], gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG)
        m.set_threads(t, arena)
        m.map(*batches[0], copy_text=False)
        t0 = time.time()
        _, h, _, _ = m.map(*batches[1], copy_text=False)
        dt = time.time() - t0
        rows.append({"threads": t, "groups": g, "arena_gb": arena / (1 << 30), "s_per_step": dt, "reads_per_s": n / dt, "gbps": n * cfg["read_len"] / dt / 1e9, "hits": len(h)})
        B.log(json.dumps(rows[-1]))
        m.close(); ctx.close()
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
