#!/usr/bin/env python
"""bench.py — mapped Gbp/s of the seed→chain→align hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL; see the contract in the task statement)

Workload (BASELINE.json configs[1]): 15 kb ONT-profile reads (3 % sub, 3 % ins, 4 % del, 1 % of
reads with one SV) against a 250 Mb synthetic reference (25 contigs x 10 Mb, 10 % repeat families incl. 171-bp
satellite arrays), `-W` list = canonical 15-mers above the 0.9998-distinct threshold, preset map-ont, CIGARs on.
A STEP = one mini-batch of reads (--reads-per-step per rank; default 65 536 x 15 kb = 0.98 Gbase, the reference's own
mini-batch size `-K 1G`, src/options.c:50) through the full path: sketch → seed → chain → ksw kernels with
the host MCAS glue in between; reads shard across ranks (weak scaling: per-GPU work is fixed), no data-path collective.
The reference index is built by rank 0 and broadcast with RCCL (torch.distributed "nccl") as flat arrays.

`value` = read bases of all ranks mapped in the timed steps / wall time, inputs resident on the host as the C-ABI takes
host buffers for reads (the PCIe-inclusive figure is therefore what is reported).
`roofline`: dominant kernel = ksw_dp (1 B of traceback per DP cell is > 99 % of the path's algorithmic bytes);
achieved = DP cells of the timed steps ÷ the kernels' summed duration (HIP events on their streams).
`cpu_baseline`: the REAL reference (oracle/_ref/winnowmap_ref, built from /root/reference) mapping a bounded sample of
the same reads against the same reference on this host with -t <all cores>; mapping-phase wall time only.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")   # before HIP initialises (see winnowmap_amd/__init__.py)
from winnowmap_amd import gpu, synth  # noqa: E402
from winnowmap_amd import dist as wmdist  # noqa: E402


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def make_workload(ref_mb, tmp):
    n_contigs = max(1, int(round(ref_mb / 10.0)))
    clen = int(ref_mb * 1e6 / n_contigs)
    t0 = time.time()
    ref = synth.make_reference(n_contigs, clen, 3, repeat_frac=0.10)
    fa = os.path.join(tmp, "ref.fa")
    synth.write_fasta(fa, ref, prefix="ctg")
    kf = os.path.join(tmp, "repetitive_k15.txt")
    n_k = gpu.write_repetitive_kmers(fa, 15, kf)
    log("reference %d x %d bp, -W list %d k-mers (%.1fs)" % (n_contigs, clen, n_k, time.time() - t0))
    return ref, fa, kf


def cpu_reference_baseline(fa, kf, reads, tmp, n_cores):
    """Map a bounded sample with the real reference binary; mapping phase = Real time − 'loaded/built the index' stamp."""
    binp = os.path.join(ROOT, "oracle", "_ref", "winnowmap_ref")
    if not os.path.exists(binp):
        return None
    rq = os.path.join(tmp, "sample.fa")
    with open(rq, "wb") as f:
        for i, s in enumerate(reads):
            f.write(b">s%d\n" % i)
            f.write(s)
            f.write(b"\n")
    t0 = time.time()
    p = subprocess.run([binp, "-t", str(n_cores), "-W", kf, "-ax", "map-ont", fa, rq], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    wall = time.time() - t0
    err = p.stderr.decode(errors="ignore")
    m_idx = re.search(r"\[M::main::([0-9.]+)\*[0-9.]+\] loaded/built the index", err)
    m_end = re.search(r"Real time: ([0-9.]+) sec", err)
    if p.returncode != 0 or not m_idx or not m_end:
        log("reference baseline failed (rc=%d)" % p.returncode)
        return None
    t_map = float(m_end.group(1)) - float(m_idx.group(1))
    bases = sum(len(s) for s in reads)
    log("reference CPU baseline: %d reads, index %.1fs, mapping %.2fs on %d threads (wall %.1fs)" % (len(reads), float(m_idx.group(1)), t_map, n_cores, wall))
    return {"value": bases / t_map / 1e9, "unit": "Gbp/s", "cores": n_cores, "kind": "reference",
            "sample": "%d x 15 kb reads of the same workload, winnowmap_ref -t %d -W -ax map-ont, mapping phase %.2f s (index build %.1f s excluded)" % (len(reads), n_cores, t_map, float(m_idx.group(1)))}


KSW_CLASS_NAMES = {0: "ksw_dps_kernel<4,...>", 4: "ksw_dps_kernel<8,...>", 8: "ksw_dps_kernel<16,...>", 12: "ksw_multi_kernel<8>", 13: "ksw_multi_kernel<16>",
                   14: "ksw_block_kernel<7,0>", 15: "ksw_generic_kernel"}


def ksw_class_name(k):
    if k >= 12:
        return KSW_CLASS_NAMES[k]
    return KSW_CLASS_NAMES[k & ~3].replace("...", "%s,%s" % ("true" if k >> 1 & 1 else "false", "true" if k & 1 else "false"))


def make_report(args, world, n_threads, elapsed, total_bases, hits, ks0, ks1):
    """The JSON line (without cpu_baseline). ks0/ks1: Mapper.kernel_stats() before/after the timed region."""
    value = total_bases / elapsed / 1e9
    # dominant kernel = the ksw class with the largest summed launch time in the timed region (HIP events on its stream)
    cls = {k: (ks1[k][0] - ks0[k][0], ks1[k][1] - ks0[k][1], ks1[k][2] - ks0[k][2]) for k in ks1}
    dom = max(cls, key=lambda k: cls[k][0])
    d_ms, d_cells, d_launch = cls[dom]
    kname = ksw_class_name(dom)
    ach = d_cells / max(d_ms, 1e-9) / 1e6       # 1 B of traceback per DP cell: bytes per ms / 1e6 = GB/s
    all_ms = sum(v[0] for v in cls.values())
    all_cells = sum(v[1] for v in cls.values())
    classes = {ksw_class_name(k): {"ms": round(v[0], 3), "cells": v[1], "launches": v[2]} for k, v in cls.items() if v[2] > 0}
    # HBM traffic per DP cell of each kernel, measured in separate rocprofv3 --pmc passes (tools/pmc_ratio.py -> profiles/)
    traffic = None
    pmc = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_bytes_per_cell.json")) as f:
            pmc = json.load(f)
        if kname in pmc["bytes_per_cell"]:
            traffic = pmc["bytes_per_cell"][kname] * d_cells / max(1, d_launch)
    except Exception:  # noqa: BLE001
        pmc = None
    return {
        "metric": "mapped Gbp/s", "value": value, "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / max(1, args.steps) * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int8", "data": "synthetic",
        "config": {"workload": "%d x %d bp ONT-profile reads per step per GPU vs %.0f Mb synthetic reference (10%% repeats), -W repetitive_k15.txt -x map-ont, CIGAR on"
                               % (args.reads_per_step, args.read_len, args.ref_mb),
                   "reads_per_step_per_gpu": args.reads_per_step, "read_len": args.read_len, "ref_mb": args.ref_mb, "host_threads": n_threads,
                   "reads_per_s": total_bases / args.read_len / elapsed, "hits": hits, "parallelism": "reads sharded over %d rank(s), index broadcast" % world},
        "roofline": {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": traffic,
                     "traffic_source": (pmc or {}).get("source") if traffic is not None else None, "classes": classes,
                     "kernel": kname, "algorithmic_bytes": "1 B traceback per DP cell (sequence bytes are < 1 %)",
                     "launches": d_launch, "avg_launch_ms": d_ms / max(1, d_launch), "cells_per_launch": d_cells / max(1, d_launch),
                     "gcups_dominant": d_cells / max(d_ms, 1e-9) / 1e6, "gcups_all_ksw_classes": all_cells / max(all_ms, 1e-9) / 1e6,
                     "note": "int8 DP is VALU-issue bound (about 33 lane-ops per cell), not HBM bound; launch durations include overlap with other streams"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads-per-step", type=int, default=int(os.environ.get("WM_BENCH_READS", 65536)))
    ap.add_argument("--ref-mb", type=float, default=float(os.environ.get("WM_BENCH_REF_MB", 250)))
    ap.add_argument("--read-len", type=int, default=15000)
    ap.add_argument("--threads", type=int, default=int(os.environ.get("WM_BENCH_THREADS", 0)))
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("WM_BENCH_CPU_SAMPLE", 4096)))
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    import torch
    dist = None
    if world > 1 or os.environ.get("WM_BENCH_FORCE_DIST"):     # (the env switch lets a 1-GPU box exercise the RCCL code path)
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for N > 1"
    n_cores = os.cpu_count() or 1
    # host threads per rank: more than ~48 per process LOSES throughput on the 2 x 128-thread host (measured), 24-48 are equivalent
    n_threads = args.threads or max(1, min(32, int(0.75 * n_cores / max(1, world))))
    tmp = tempfile.mkdtemp(prefix="wmbench_")

    # ---- reference + index: rank 0 builds, RCCL broadcasts the flat arrays ----
    t0 = time.time()
    if rank == 0:
        ref, fa, kf = make_workload(args.ref_mb, tmp)
        idx = gpu.Index(fa, kf, k=15, w=50, n_threads=min(64, n_cores))
        log("index: %d minimizers (%.1fs)" % (idx.n_minimizers, time.time() - t0))
    if dist is not None:
        dev = torch.device("cuda", local)
        idx = wmdist.broadcast_index(idx if rank == 0 else None, rank, dist, dev)      # RCCL over xGMI, one broadcast per flat array
        # every rank regenerates the (seeded) reference only to draw its reads from it
        if rank != 0:
            n_contigs = max(1, int(round(args.ref_mb / 10.0)))
            ref = synth.make_reference(n_contigs, int(args.ref_mb * 1e6 / n_contigs), 3, repeat_frac=0.10)
    arena = int(os.environ.get("WM_BENCH_ARENA_GB", 48)) << 30   # per group; 288 GB of HBM per GPU
    ctx = gpu.Context(local, arena)
    idx.upload(ctx)
    mapper = gpu.Mapper(ctx, idx, "map-ont", gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG)
    if n_threads > 1:
        mapper.set_threads(n_threads, arena)

    # ---- reads: a fresh batch per step, generated before the clock starts ----
    t1 = time.time()
    n_steps = args.warmup + args.steps
    n_distinct = min(n_steps, int(os.environ.get("WM_BENCH_DISTINCT_BATCHES", 4)))   # later steps cycle through these (mapping is stateless)
    reads, _ = synth.make_reads(ref, n_distinct * args.reads_per_step, args.read_len, 4 + 1000 * rank, profile="ont", sv_frac=0.01)
    seqs = [synth.codes_to_ascii(r) for r in reads]
    del reads
    names = [("r%d_%d" % (rank, i)).encode() for i in range(len(seqs))]
    distinct = [(names[i * args.reads_per_step:(i + 1) * args.reads_per_step], seqs[i * args.reads_per_step:(i + 1) * args.reads_per_step]) for i in range(n_distinct)]
    batches = [distinct[i % n_distinct] for i in range(n_steps)]
    log("rank %d: %d reads generated (%.1fs), %d host threads" % (rank, len(seqs), time.time() - t1, n_threads))

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for b in batches[:args.warmup]:
        mapper.map(*b, copy_text=False)
    sync()
    ks0 = mapper.kernel_stats()
    t_start = time.time()
    cells = ksw_us = aux_us = bases = hits = 0
    for b in batches[args.warmup:]:
        text_len, h, _, _ = mapper.map(*b, copy_text=False)       # the records stay in the library's buffer (no Python copy)
        st = mapper.stats()
        cells += st["dp_cells"]; ksw_us += st["ksw_kernel_us"]; aux_us += st["aux_kernel_us"]; bases += st["read_bases"]; hits += len(h)
    sync()
    elapsed = time.time() - t_start
    ks1 = mapper.kernel_stats()
    if dist is not None:
        dev = torch.device("cuda", local)
        elapsed = wmdist.max_over_ranks(elapsed, dist, dev)
        total_bases = wmdist.sum_over_ranks([bases], dist, dev)[0]
    else:
        total_bases = float(bases)

    if rank == 0:
        out = make_report(args, world, n_threads, elapsed, total_bases, hits, ks0, ks1)
        if world == 1 and args.cpu_sample > 0:
            sample = [s for _, ss in batches[args.warmup:] for s in ss][:args.cpu_sample]
            try:
                out["cpu_baseline"] = cpu_reference_baseline(fa, kf, sample, tmp, n_cores)
            except Exception as e:  # noqa: BLE001
                log("cpu baseline error:", e)
                out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
