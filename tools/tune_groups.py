#!/usr/bin/env python
"""One-process sweep of (host threads, device contexts[, reads in flight]) on the bench workload: reads/s per configuration.
  python tools/tune_groups.py "16:4,16:8:32768,16:6:16384:WM_NO_RESIDENT=1;WM_KSW_HEAVY_UNITS=0" [--config 2] [--reads 65536]"""
import argparse
import importlib.util
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
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
        parts = spec_.split(":")
        envs = dict(kv.split("=", 1) for kv in parts[3].split(";")) if len(parts) > 3 and parts[3] else {}
        f = [int(x) for x in parts[:3]]
        t, g, infl = f[0], f[1], (f[2] if len(f) > 2 else 0)
        for k_, v_ in envs.items():
            os.environ[k_] = v_
        os.environ["WM_GROUPS"] = str(g)
        os.environ["WM_INFLIGHT"] = str(infl) if infl else ""
        if not infl:
            os.environ.pop("WM_INFLIGHT")
        arena = int(min(48.0, a.hbm_gb / g) * (1 << 30))
        ctx = gpu.Context(0, arena)
        idx.upload(ctx)
        m = gpu.Mapper(ctx, idx, cfg["preset"], gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG)
        m.set_threads(t, arena)
        m.map(batches[0][0][:n // 4], batches[0][1][:n // 4], copy_text=False)     # warm-up (pinned slabs, code objects)
        hs0 = m.host_stats()
        t0 = time.time()
        _, h, _, _ = m.map(*batches[1], copy_text=False)
        dt = time.time() - t0
        hs1 = m.host_stats()
        for k_ in envs:
            os.environ.pop(k_)
        rows.append({"env": envs, "threads": t, "contexts": g, "inflight": infl or 16384, "arena_gb": arena / (1 << 30), "s_per_step": round(dt, 3), "reads_per_s": round(n / dt, 1), "gbps": round(n * cfg["read_len"] / dt / 1e9, 4), "hits": len(h),
                     "glue_cpu_s": round(hs1["cpu_glue_s"] - hs0["cpu_glue_s"], 1), "help_cpu_s": round(hs1["cpu_help_s"] - hs0["cpu_help_s"], 1), "idle_wall_s": round(hs1["idle_wall_s"] - hs0["idle_wall_s"], 1),
                     "batched_wall_s": {o: round(hs1["wall_batched_s"][o] - hs0["wall_batched_s"][o], 1) for o in hs1["wall_batched_s"]},
                     "batched_calls": {o: hs1["batched_calls"][o] - hs0["batched_calls"][o] for o in hs1["batched_calls"]}})
        B.log(json.dumps(rows[-1]))
        m.close(); ctx.close()
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
