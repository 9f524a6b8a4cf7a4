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
The `-W` list is counted and the reference index is built by rank 0 ON ITS GPU (wm_write_repetitive_kmers_gpu, wm_index_build_gpu: one
wavefront per contig sketches the reference) and broadcast with RCCL (torch.distributed "nccl") as flat arrays — outside the timed region.

`value` = read bases of all ranks mapped in the timed steps / wall time, inputs resident on the host as the C-ABI takes
host buffers for reads (the PCIe-inclusive figure is therefore what is reported). Inside a step the read codes
are uploaded once and every alignment / sketch request refers to positions in them and in the packed reference.
`host`: where the host time of the timed region went (per-read glue, batched calls, helpers, idle) against the usable cores.
`roofline`: dominant kernel = ksw_dp (1 B of traceback per DP cell is > 99 % of the path's algorithmic bytes);
achieved = DP cells of the timed steps ÷ the kernels' summed duration (HIP events on their streams).
`cpu_baseline`: the REAL reference (oracle/_ref/winnowmap_ref, built from /root/reference) mapping ONE FULL STEP of the
same reads against the same reference on this host, swept over thread counts; the BEST mapping-phase rate is reported.
`parity`: the reference's PAF records for that step are diffed against ours for the same reads (every column and tag
except MAPQ / rl:i, see winnowmap_amd/parity.py); outside the timed region.
`--config 3` switches to BASELINE config 3 (50 000 x 20 kb HiFi reads, map-pb); config 2 stays the default.
`--gpus N` without a torch.distributed environment re-launches itself under torch.distributed.run with N ranks.
"""
import argparse
import json
import os
import re
import resource
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")   # before HIP initialises (see winnowmap_amd/__init__.py)
from winnowmap_amd import gpu, synth  # noqa: E402
from winnowmap_amd import dist as wmdist  # noqa: E402
from winnowmap_amd import parity as wmparity  # noqa: E402


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def make_workload(ref_mb, tmp, ctx=None):
    """the synthetic reference and its -W list (counted on the device when a context is given, on the host otherwise: same file)"""
    n_contigs = max(1, int(round(ref_mb / 10.0)))
    clen = int(ref_mb * 1e6 / n_contigs)
    t0 = time.time()
    ref = synth.make_reference(n_contigs, clen, 3, repeat_frac=0.10)
    fa = os.path.join(tmp, "ref.fa")
    synth.write_fasta(fa, ref, prefix="ctg")
    kf = os.path.join(tmp, "repetitive_k15.txt")
    t1 = time.time()
    if ctx is not None:
        n_k, st = gpu.write_repetitive_kmers_gpu(ctx, fa, 15, kf)
    else:
        n_k = gpu.write_repetitive_kmers(fa, 15, kf)
    log("reference %d x %d bp (%.1fs), -W list %d k-mers (%.1fs on the %s)" % (n_contigs, clen, t1 - t0, n_k, time.time() - t1, "device" if ctx is not None else "host"))
    return ref, fa, kf


CONFIGS = {
    # BASELINE.json configs[1] / configs[2]: reads per step = the reference's 1-Gbase mini-batch (-K 1G)
    2: {"preset": "map-ont", "profile": "ont", "read_len": 15000, "reads_per_step": 65536, "sv_frac": 0.01, "seed": 4, "label": "ONT-profile"},
    3: {"preset": "map-pb", "profile": "hifi", "read_len": 20000, "reads_per_step": 50000, "sv_frac": 0.0, "seed": 5, "label": "HiFi-profile"},
    # BASELINE.json configs[3] on ONE GPU (one rank's share of "1M x 15 kb vs 3 Gb, sharded across 8"): the same reads against a 3-Gbase reference — the index
    # (~1.2 x 10^8 minimizers, a 2^28-slot table) no longer sits in L2, which is what changes for the seed stage (src/index.c:88-105)
    4: {"preset": "map-ont", "profile": "ont", "read_len": 15000, "reads_per_step": 65536, "sv_frac": 0.01, "seed": 4, "label": "ONT-profile", "ref_mb": 3000.0},
}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def run_reference(fa, kf, rq, preset, n_threads, out_path, sam=False):
    """One run of the real reference binary; -> (mapping-phase seconds, index seconds, wall seconds) or None.
    Mapping phase = 'Real time' (src/main.c:441) − the 'loaded/built the index' stamp (src/main.c:401)."""
    binp = os.path.join(ROOT, "oracle", "_ref", "winnowmap_ref")
    cmd = [binp, "-t", str(n_threads), "-W", kf, "-ax" if sam else "-cx", preset, fa, rq]
    t0 = time.time()
    with open(out_path, "wb") as fo:
        p = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE)
    wall = time.time() - t0
    err = p.stderr.decode(errors="ignore")
    m_idx = re.search(r"\[M::main::([0-9.]+)\*[0-9.]+\] loaded/built the index", err)
    m_end = re.search(r"Real time: ([0-9.]+) sec", err)
    if p.returncode != 0 or not m_idx or not m_end:
        log("reference run failed (rc=%d): %s" % (p.returncode, err[-300:]))
        return None
    return float(m_end.group(1)) - float(m_idx.group(1)), float(m_idx.group(1)), wall


def write_reads_fasta(path, names, seqs):
    with open(path, "wb") as f:
        for n, s in zip(names, seqs):
            f.write(b">" + n + b"\n")
            f.write(s)
            f.write(b"\n")


def cpu_thread_candidates(n_cores):
    """Thread counts of the baseline sweep: WM_BENCH_CPU_THREADS="16,32" overrides; default = the usable cores (container quota) and
    twice that — the two best points of the full sweep on the GPU boxes (profiles/r02_cpu_sweep_c2*.txt: the reference is fastest
    at -t 16..32 under the 16-CPU quota and slows down steadily beyond)."""
    env = os.environ.get("WM_BENCH_CPU_THREADS")
    if env:
        return [int(x) for x in env.split(",") if x]
    return sorted({n_cores, 2 * n_cores})


def cpu_reference_baseline(fa, kf, names, seqs, preset, tmp, n_cores):
    """Map one full step with the real reference binary (PAF + CIGAR, like the timed GPU path) at each candidate thread
    count; report the BEST mapping-phase rate. Returns (cpu_baseline dict, path of the reference's PAF)."""
    binp = os.path.join(ROOT, "oracle", "_ref", "winnowmap_ref")
    if not os.path.exists(binp):
        return None, None
    rq = os.path.join(tmp, "step.fa")
    write_reads_fasta(rq, names, seqs)
    bases = sum(len(s) for s in seqs)
    best = None
    table = []
    for t in cpu_thread_candidates(n_cores):
        outp = os.path.join(tmp, "ref_t%d.paf" % t)
        r = run_reference(fa, kf, rq, preset, t, outp)
        if r is None:
            continue
        table.append({"threads": t, "map_s": round(r[0], 3), "gbps": bases / r[0] / 1e9})
        log("reference CPU baseline: %d reads, -t %d: index %.1fs, mapping %.2fs (wall %.1fs) = %.4f Gbp/s" % (len(seqs), t, r[1], r[0], r[2], bases / r[0] / 1e9))
        if best is None or r[0] < best[0]:
            best = (r[0], t, outp, r[1])
    if best is None:
        return None, None
    t_map, t_best, outp, t_idx = best
    return {"value": bases / t_map / 1e9, "unit": "Gbp/s", "cores": t_best, "kind": "reference", "cpu": cpu_model(), "hardware_threads": os.cpu_count(), "usable_cores": n_cores, "sweep": table,
            "sample": "one full step (%d reads, %.2f Gbase) of the same workload, winnowmap_ref -t %d -W -cx %s (PAF+CIGAR as in the timed GPU path), "
                      "mapping phase %.2f s = best of the -t sweep (index build %.1f s excluded)" % (len(seqs), bases / 1e9, t_best, preset, t_map, t_idx)}, outp


KSW_WIDE_CLASSES = {24: "ksw_pmulti_kernel<4, 8>", 25: "ksw_pmulti_kernel<8, 8>", 26: "ksw_block_kernel<7, 0>", 27: "ksw_generic_kernel"}


def ksw_class_name(k):
    """kernel name of a ksw class id (winnowmap_amd/csrc/ksw_plan.h: window*8 + EXACT*4 + CLIP*2 + HASN; 24.. = the wide-hull kernels);
    spelled as rocprofv3 prints the instantiation ksw_dpp_kernel<BP, CLIP, HASN, EXACT> (jobs with an N run on the CLIP instantiation)"""
    if k >= 52:                                                    # chained-workgroup classes: WM_KSW_CHAIN + geometry * 4 + CLIP * 2 + HASN
        g, v = (k - 52) >> 2, (k - 52) & 3
        return "ksw_chain_kernel<%s, %s, %s, *>" % ((2, 4)[g], str(bool(v & 2) or bool(v & 1)).lower(), str(bool(v & 1)).lower())      # (two kernels per class: <.., EXACT>)
    if k >= 28:                                                    # stripe classes: WM_KSW_STRIPE + geometry * 4 + CLIP * 2 + HASN
        g, v = (k - 28) >> 2, (k - 28) & 3
        return "ksw_stripe_kernel<%s, %s, %s>" % (("2, 4", "2, 8", "4, 8", "8, 8", "1, 16", "2, 16")[g], str(bool(v & 2) or bool(v & 1)).lower(), str(bool(v & 1)).lower())
    if k >= 24:
        return KSW_WIDE_CLASSES[k]
    if k >= 16 and int(os.environ.get("WM_KSW_PMULTI", 2)) >= 2:       # the 16-pair classes run on 4 wavefronts per alignment (library default)
        return "ksw_pmulti_kernel<4, 4>"
    bp = (4, 8, 16)[k >> 3]
    exact, clip, hasn = bool(k & 4), bool(k & 2) or bool(k & 1), bool(k & 1)
    if (k & 7) == 0 and bp <= 8 and os.environ.get("WM_KSW_DUAL", "0") not in ("", "0"):      # the gap-fill classes with two alignments per wavefront (ksw_dual_kernel.h, round 6; off by default)
        return "ksw_dual_kernel<%d>" % (2 * bp)
    return "ksw_dpp_kernel<%d, %s, %s, %s>" % (bp, str(clip).lower(), str(hasn).lower(), str(exact).lower())


def cgroup_throttle():
    """(periods throttled, seconds throttled) of this container's CPU quota so far (cgroup v2 cpu.stat; zeros when unreadable)"""
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            kv = dict(l.split() for l in open(path).read().splitlines() if len(l.split()) == 2)
            return int(kv.get("nr_throttled", 0)), float(kv.get("throttled_usec", kv.get("throttled_time", 0))) * (1e-6 if "throttled_usec" in kv else 1e-9)
        except (OSError, ValueError):
            continue
    return 0, 0.0


def host_report(hs0, hs1, ru0, ru1, elapsed, n_cores):
    """Where the host time of the timed region went (rank 0): the per-read glue and the batched calls, against the usable cores."""
    d = lambda a, b: b - a  # noqa: E731
    cpu = d(ru0.ru_utime, ru1.ru_utime) + d(ru0.ru_stime, ru1.ru_stime)
    ops = ("window", "ksw")
    return {"usable_cores": n_cores, "process_cpu_s": round(cpu, 2), "cpu_utilisation": round(cpu / max(elapsed, 1e-9) / max(1, n_cores), 3),
            "glue_wall_s": round(d(hs0["glue_wall_s"], hs1["glue_wall_s"]), 2), "lock_wait_wall_s": round(d(hs0["lock_wait_wall_s"], hs1["lock_wait_wall_s"]), 2),
            "workers_wall_s": round(d(hs0["workers_wall_s"], hs1["workers_wall_s"]), 2),
            "glue_cpu_s": round(d(hs0["cpu_glue_s"], hs1["cpu_glue_s"]), 2), "help_cpu_s": round(d(hs0["cpu_help_s"], hs1["cpu_help_s"]), 2), "idle_wall_s": round(d(hs0["idle_wall_s"], hs1["idle_wall_s"]), 2),
            "batched_cpu_s": {o: round(d(hs0["cpu_batched_s"][o], hs1["cpu_batched_s"][o]), 2) for o in ops},
            "batched_wall_s": {o: round(d(hs0["wall_batched_s"][o], hs1["wall_batched_s"][o]), 2) for o in ops},
            "batched_calls": {o: hs1["batched_calls"][o] - hs0["batched_calls"][o] for o in ops},
            "map_wall_s": round(d(hs0["map_wall_s"], hs1["map_wall_s"]), 2), "format_wall_s": round(d(hs0["format_wall_s"], hs1["format_wall_s"]), 2)}


VALU_CEILING = {"lane_ops_per_s": 37.4e12, "wave_instr_per_s": 37.4e12 / 64, "packed_valu_per_128_cells": 53,
                "gcups": 37.4e12 / 64 / 53 * 128 / 1e9, "source": "tools/ubench/valu_bench.hip on MI355X: profiles/r02_valu_bench.txt (packed 16-bit VALU, all CUs)"}


def make_report(args, world, n_threads, elapsed, total_bases, hits, ks0, ks1, union_ms=None):
    """The JSON line (without cpu_baseline). ks0/ks1: Mapper.kernel_stats() before/after the timed region; union_ms: per class id, the ms with at
    least one launch of the class running inside the timed region (Mapper.kernel_union)."""
    value = total_bases / elapsed / 1e9
    # dominant kernel = the ksw kernel (instantiation) with the largest summed launch time in the timed region (HIP events on its stream);
    # several job classes can share one kernel (the 16-pair classes all run on ksw_pmulti_kernel<4, 4>)
    cls = {}
    for k in ks1:
        v = (ks1[k][0] - ks0[k][0], ks1[k][1] - ks0[k][1], ks1[k][2] - ks0[k][2])
        o = cls.get(ksw_class_name(k), (0.0, 0.0, 0))
        cls[ksw_class_name(k)] = (o[0] + v[0], o[1] + v[1], o[2] + v[2])
    kname = max(cls, key=lambda k: cls[k][0])
    d_ms, d_cells, d_launch = cls[kname]
    ach = d_cells / max(d_ms, 1e-9) / 1e6       # 1 B of traceback per DP cell: bytes per ms / 1e6 = GB/s
    all_ms = sum(v[0] for v in cls.values())
    all_cells = sum(v[1] for v in cls.values())
    classes = {k: {"ms": round(v[0], 3), "cells": v[1], "launches": v[2]} for k, v in cls.items() if v[2] > 0}
    # launches of one kernel overlap on different streams: `ms` above is summed residency; `union_ms` is the time with at least one launch running
    if union_ms:
        un = {}
        for k, v in union_ms.items():
            if k in ks1:
                un[ksw_class_name(k)] = un.get(ksw_class_name(k), 0.0) + v      # (classes sharing a kernel: an upper bound of their union)
        for k in classes:
            classes[k]["union_ms"] = round(un.get(k, 0.0), 3)
            classes[k]["gcups_residency"] = round(classes[k]["cells"] / max(classes[k]["ms"], 1e-9) / 1e6, 2)
            classes[k]["gcups_union"] = round(classes[k]["cells"] / max(classes[k]["union_ms"], 1e-9) / 1e6, 2) if classes[k]["union_ms"] > 0 else None
    # instructions per DP cell of each kernel, from separate rocprofv3 --pmc SQ_INSTS_VALU / SQ_INSTS_SALU passes (tools/pmc_insts.py -> profiles/)
    ipc = None
    try:
        with open(os.path.join(ROOT, "profiles", "insts_per_cell.json")) as f:
            ipc = json.load(f)
        for k in classes:
            if k in ipc.get("kernels", {}):
                classes[k].update({"valu_per_cell": ipc["kernels"][k].get("valu_per_cell"), "salu_per_cell": ipc["kernels"][k].get("salu_per_cell")})
    except Exception:  # noqa: BLE001
        ipc = None
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
        "config": {"workload": "BASELINE config %d: %d x %d bp %s reads per step per GPU vs %.0f Mb synthetic reference (10%% repeats), -W repetitive_k15.txt -x %s, CIGAR on"
                               % (args.config, args.reads_per_step, args.read_len, CONFIGS[args.config]["label"], args.ref_mb, CONFIGS[args.config]["preset"]),
                   "reads_per_step_per_gpu": args.reads_per_step, "read_len": args.read_len, "ref_mb": args.ref_mb, "host_threads": n_threads,
                   "reads_per_s": total_bases / args.read_len / elapsed, "hits": hits, "parallelism": "reads sharded over %d rank(s), index broadcast" % world, "mini_batches_in_flight": max(1, min(4, int(os.environ.get("WM_BENCH_SLOTS", 2)))) if int(os.environ.get("WM_BENCH_PIPELINE", 1)) else 1,
                   # which kernel variants ran (tools/r03_first_run.sh A/Bs them): build-time defines and run-time switches
                   "variants": {"kernel_defines": gpu.build_defines(),
                                **{k: os.environ[k] for k in ("WM_KSW_PMULTI", "WM_KSW_COOP_BT", "WM_SEED_DEVICE_SORT", "WM_CONTEXTS") if k in os.environ}}},
        "roofline": {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": traffic,
                     "traffic_source": (pmc or {}).get("source") if traffic is not None else None, "classes": classes,
                     "kernel": kname, "algorithmic_bytes": "1 B traceback per DP cell (sequence bytes are < 1 %)",
                     "launches": d_launch, "avg_launch_ms": d_ms / max(1, d_launch), "cells_per_launch": d_cells / max(1, d_launch),
                     "gcups_dominant": d_cells / max(d_ms, 1e-9) / 1e6, "gcups_all_ksw_classes": all_cells / max(all_ms, 1e-9) / 1e6,
                     # what bounds the kernels is instruction issue, not memory: the measured packed-VALU ceiling of the chip, the DP rate it allows at the
                     # 53 packed VALU instructions per 128 cells of the lean register kernel, and how far the step and the dominant kernel are from it
                     "valu_ceiling": VALU_CEILING,
                     "gcups_step": all_cells / max(elapsed, 1e-9) / 1e9 / max(1, world),
                     "issue_frac_step": all_cells / max(elapsed, 1e-9) / 1e9 / max(1, world) / VALU_CEILING["gcups"],
                     "issue_frac_dominant": d_cells / max(d_ms, 1e-9) / 1e6 / VALU_CEILING["gcups"],
                     # the dominant kernel's MEASURED instruction mix next to the 53 packed VALU per 128 cells the ceiling assumes: the distance of
                     # issue_frac_dominant from 1 is partly instruction count (this ratio) and partly latency (a lone wavefront per SIMD issues one per ~10 cycles)
                     "dominant_valu_salu_per_128_cells": ([round(128 * classes[kname]["valu_per_cell"], 1), round(128 * classes[kname]["salu_per_cell"], 1)]
                                                          if classes.get(kname, {}).get("valu_per_cell") is not None and classes[kname].get("salu_per_cell") is not None else None),
                     "insts_per_cell_source": (ipc or {}).get("source"),
                     "note": "int8 DP is instruction-issue bound, not HBM bound (frac above is the HBM fraction of the 1 B/cell traceback stream); `ms` of a class is the SUM of its launch "
                             "durations (launches overlap on different streams: residency), `union_ms` the time with at least one launch running; gcups_step = cells of all classes / wall time of the timed region"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)       # (an even number: two mini-batches are in flight, an odd step count ends with one lane alone)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=int(os.environ.get("WM_BENCH_CONFIG", 2)), choices=sorted(CONFIGS))
    ap.add_argument("--reads-per-step", type=int, default=int(os.environ.get("WM_BENCH_READS", 0)))
    ap.add_argument("--ref-mb", type=float, default=float(os.environ.get("WM_BENCH_REF_MB", 0)))
    ap.add_argument("--read-len", type=int, default=0)
    ap.add_argument("--threads", type=int, default=int(os.environ.get("WM_BENCH_THREADS", 0)))
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("WM_BENCH_CPU_SAMPLE", -1)),
                    help="reads of the CPU baseline / parity leg: -1 = one full step, 0 = skip the leg")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    args.reads_per_step = args.reads_per_step or cfg["reads_per_step"]
    args.read_len = args.read_len or cfg["read_len"]
    args.ref_mb = args.ref_mb or cfg.get("ref_mb", 250.0)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher of N ranks (one per GPU) — never run 1 rank and call it N
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit("bench.py: --gpus %d requested but %d GPU(s) are visible on this box" % (args.gpus, have))
        sys.exit(wmdist.launch_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, or plainly and let bench.py launch the ranks)" % (args.gpus, world, args.gpus))
    import torch
    dist = None
    if world > 1 or os.environ.get("WM_BENCH_FORCE_DIST"):     # (the env switch lets a 1-GPU box exercise the RCCL code path)
        import torch.distributed as dist
        if torch.cuda.device_count() <= local:
            sys.exit("bench.py: rank %d has no GPU %d (%d visible)" % (rank, local, torch.cuda.device_count()))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        assert dist.get_world_size() == args.gpus and dist.get_backend() == "nccl"
    n_cores = wmdist.available_cores()       # hardware threads, affinity and the container CPU quota
    # host threads per rank: the ranks of one node share its cores — divide them explicitly
    n_threads = args.threads or wmdist.host_threads_per_rank(n_cores, world)
    if rank == 0 and world > 1 and n_threads < 8:
        log("WARNING: %d usable host cores shared by %d ranks = %d host threads per GPU; the per-read glue needs about %s CPU-s per Gbase (see the `host` block of a 1-GPU run), "
            "so this node's host side, not its GPUs, bounds the scaling curve" % (n_cores, world, n_threads, "35"))
    tmp = tempfile.mkdtemp(prefix="wmbench_")

    # ---- reference + index: rank 0 builds, RCCL broadcasts the flat arrays ----
    t0 = time.time()
    # device contexts (own streams + arena each: that many batched calls are on the device at once) share the 288 GB of HBM
    n_ctx = int(os.environ.setdefault("WM_CONTEXTS", "6"))
    arena = int(float(os.environ.get("WM_BENCH_ARENA_GB", min(48.0, 216.0 / n_ctx))) * (1 << 30))
    ctx = gpu.Context(local, arena)
    if rank == 0:
        ref, fa, kf = make_workload(args.ref_mb, tmp, ctx)
        t_i = time.time()
        idx, ist = gpu.Index.build_on_device(ctx, fa, kf, k=15, w=50, n_threads=min(64, n_cores))      # map-ont and map-pb: k=15, w=50 (src/options.c:94-103)
        log("index: %d minimizers, built in %.1fs (reference sketched on the device in %.2fs)" % (idx.n_minimizers, time.time() - t_i, ist["device_sketch_s"]))
    if dist is not None:
        dev = torch.device("cuda", local)
        idx, uploaded = wmdist.broadcast_index(idx if rank == 0 else None, rank, dist, dev, ctx)      # RCCL over xGMI, one broadcast per flat array, received straight into the context
        # every rank regenerates the (seeded) reference only to draw its reads from it
        if rank != 0:
            n_contigs = max(1, int(round(args.ref_mb / 10.0)))
            ref = synth.make_reference(n_contigs, int(args.ref_mb * 1e6 / n_contigs), 3, repeat_frac=0.10)
    if dist is None or not uploaded:
        idx.upload(ctx)
    os.environ.setdefault("WM_READ_SLABS", str(max(1, min(4, int(os.environ.get("WM_BENCH_SLOTS", 2))))))      # resident read slabs = mini-batches in flight (below)
    mapper = gpu.Mapper(ctx, idx, cfg["preset"], gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG)
    if n_threads > 1:
        mapper.set_threads(n_threads, arena)

    # ---- reads: a fresh batch per step, generated before the clock starts ----
    t1 = time.time()
    n_steps = args.warmup + args.steps
    # later steps cycle through these (mapping is stateless). Every distinct batch is mapped once before the timed region when there is a warm-up:
    # the library's pinned pools and result buffers grow to the largest batch they have seen, and a batch first seen inside the timed region
    # pays for that growth (0.2545 vs 0.272 Gbp/s, profiles/r03o_*)
    # (the cap by the warm-up count made every timed step a batch the library had already seen. Now up to 4 distinct batches whatever
    # the warm-up; the JSON line says how many timed steps ran on batches first seen inside the timed region — all batches have the same shape, so
    # what a first sight costs is the growth of pinned pools / result buffers to this batch's hit count)
    n_distinct = min(n_steps, int(os.environ.get("WM_BENCH_DISTINCT_BATCHES", 4)))
    # WM_BENCH_CACHE=<dir>: A/B runs of one GPU call share the generated reads (same seeds = same reads; only the generation time is saved)
    cache = os.environ.get("WM_BENCH_CACHE")
    cpath = os.path.join(cache, "reads_c%d_r%d_n%d_l%d_m%d.npz" % (args.config, rank, n_distinct * args.reads_per_step, args.read_len, int(args.ref_mb))) if cache else None
    if cpath and os.path.exists(cpath):
        z = np.load(cpath)
        blob, ends = z["blob"].tobytes(), z["ends"]
        seqs = [blob[(ends[i - 1] if i else 0):ends[i]] for i in range(len(ends))]
        del blob
    else:
        reads, _ = synth.make_reads(ref, n_distinct * args.reads_per_step, args.read_len, cfg["seed"] + 1000 * rank, profile=cfg["profile"], sv_frac=cfg["sv_frac"])
        seqs = [synth.codes_to_ascii(r) for r in reads]
        del reads
        if cpath:
            os.makedirs(cache, exist_ok=True)
            np.savez(cpath, blob=np.frombuffer(b"".join(seqs), np.uint8), ends=np.cumsum([len(x) for x in seqs]))
    names = [("r%d_%d" % (rank, i)).encode() for i in range(len(seqs))]
    distinct = [(names[i * args.reads_per_step:(i + 1) * args.reads_per_step], seqs[i * args.reads_per_step:(i + 1) * args.reads_per_step]) for i in range(n_distinct)]
    marshalled = [gpu.Mapper.marshal(*d) for d in distinct]       # the C argument arrays of wm_map_reads, built once (not the mapper's work)
    batches = [distinct[i % n_distinct] for i in range(n_steps)]
    mbatches = [marshalled[i % n_distinct] for i in range(n_steps)]
    log("rank %d: %d reads generated (%.1fs), %d host threads" % (rank, len(seqs), time.time() - t1, n_threads))

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Consecutive steps (mini-batches) are mapped on the mapper's two slots by two host threads (wm_map_reads_slot): a mapping call spends its
    # first and last few hundred milliseconds filling and draining its pipeline of dependent device calls, and with two mini-batches in flight those
    # phases of one hide behind the steady state of the other. Every step is one mini-batch, mapped completely inside the timed region; the
    # steps of a slot run in order. WM_BENCH_PIPELINE=0: one step at a time (A/B).
    import threading
    n_slots = max(1, min(4, int(os.environ.get("WM_BENCH_SLOTS", 2)))) if int(os.environ.get("WM_BENCH_PIPELINE", 1)) else 1

    def run_steps(step_batches):
        """maps the given mini-batches on n_slots lanes (each lane takes the next unmapped step, like wm_map_file's lanes); returns (hits, bases)"""
        tot = [0] * n_slots
        err = []
        nxt = [0]
        lock = threading.Lock()

        def worker(s):
            try:
                while True:
                    with lock:
                        i = nxt[0]
                        nxt[0] += 1
                    if i >= len(step_batches):
                        break
                    _, h, _, _ = mapper.map(step_batches[i], copy_text=False, slot=s)        # the records stay in the library's buffer (no Python copy)
                    tot[s] += len(h)
            except Exception as e:  # noqa: BLE001
                err.append(e)
        th = [threading.Thread(target=worker, args=(s,)) for s in range(n_slots)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if err:
            raise err[0]
        return sum(tot), sum(int(b[3].sum()) for b in step_batches)

    run_steps(mbatches[:args.warmup])
    sync()
    seen = set(i % n_distinct for i in range(args.warmup))
    unseen_steps = 0
    for i in range(args.warmup, n_steps):
        if i % n_distinct not in seen:
            unseen_steps += 1
            seen.add(i % n_distinct)
    _, dev_t0 = mapper.kernel_union(0.0)
    ks0 = mapper.kernel_stats()
    hs0 = mapper.host_stats()
    thr0 = cgroup_throttle()
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    if os.environ.get("SPROF_MARK"):        # tools/sprof: keep only the samples of the timed region and what follows it
        import signal
        os.kill(os.getpid(), signal.SIGUSR2)
    t_start = time.time()
    hits, bases = run_steps(mbatches[args.warmup:])
    sync()
    elapsed = time.time() - t_start
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    ks1 = mapper.kernel_stats()
    hs1 = mapper.host_stats()
    if dist is not None:
        dev = torch.device("cuda", local)
        elapsed = wmdist.max_over_ranks(elapsed, dist, dev)
        total_bases = wmdist.sum_over_ranks([bases], dist, dev)[0]
    else:
        total_bases = float(bases)

    if rank == 0:
        union_ms, _ = mapper.kernel_union(dev_t0)
        out = make_report(args, world, n_threads, elapsed, total_bases, hits, ks0, ks1, union_ms)
        out["config"]["distinct_batches"] = n_distinct
        out["config"]["timed_steps_on_batches_first_seen_in_the_timed_region"] = unseen_steps
        if dist is not None:
            out["config"]["rccl_world_size"] = dist.get_world_size()
        out["host"] = host_report(hs0, hs1, ru0, ru1, elapsed, n_cores)
        # what the host side alone allows: read bases per CPU-second spent (all threads of this rank) x the cores this rank may use
        cpu_per_base = out["host"]["process_cpu_s"] / max(1.0, float(bases))
        out["host"]["host_threads"] = n_threads
        out["host"]["host_bound_gbps_per_rank"] = round(min(n_threads, n_cores) / max(cpu_per_base, 1e-12) / 1e9, 4)
        out["host"]["host_bound_note"] = "bases per CPU-second of this rank x its host threads: the rate at which the host glue alone could feed one GPU; with N ranks on one node the usable cores are divided by N"
        thr1 = cgroup_throttle()
        out["host"]["cpu_quota_throttled"] = {"periods": thr1[0] - thr0[0], "seconds": round(thr1[1] - thr0[1], 3)}
        # what THIS node's host side allows at 8 ranks: all usable cores / the CPU-seconds one base costs — next to 8 x this rank's rate
        out["host"]["host_bound_gbps_node"] = round(n_cores / max(cpu_per_base, 1e-12) / 1e9, 4)
        out["host"]["predicted_8rank_gbps"] = round(min(8 * out["value"] / max(1, world), n_cores / max(cpu_per_base, 1e-12) / 1e9), 4)
        out["host"]["predicted_8rank_note"] = "min(8 x the per-rank rate of this run, usable cores / CPU-s per base): with %d usable cores the host side caps an 8-rank node here; not a measurement" % n_cores
        if world == 1 and int(os.environ.get("WM_BENCH_FILE", 1)):
            # the drop-in path: reads.fa -> out.paf through wm_map_file (reader / two mapping lanes / ordered writer, host/wm_pipeline.cpp) on the
            # mini-batches of the timed region: FASTA parsing and record output inside the clock, as in the reference's mapping phase (src/map.c:1107-1224)
            try:
                nb = min(2, len(distinct))
                rq_f = os.path.join(tmp, "f2f_reads.fa")
                fn, fs_ = [], []
                for d_ in distinct[:nb]:
                    fn += d_[0]; fs_ += d_[1]
                write_reads_fasta(rq_f, fn, fs_)
                fbases = sum(len(x) for x in fs_)
                t_f = time.time()
                st_f = mapper.map_file(rq_f, os.path.join(tmp, "f2f_out.paf"), int(fbases // nb) + 1)
                wall_f = time.time() - t_f
                out["file_to_file"] = {"value": fbases / wall_f / 1e9, "unit": "Gbp/s", "reads": len(fs_), "mini_batches": int(st_f["batches"]), "wall_s": round(wall_f, 3),
                                       "read_s": round(st_f["t_read"], 3), "map_s": round(st_f["t_map"], 3), "write_s": round(st_f["t_write"], 3),
                                       "paf_bytes": os.path.getsize(os.path.join(tmp, "f2f_out.paf")),
                                       "note": "wm_map_file(reads.fa -> out.paf): %d mini-batch(es) of one step each, FASTA parsing and PAF output inside the clock; reported beside `value`, not instead" % nb}
                os.unlink(rq_f)
                log("file to file: %d reads, %.2f s = %.4f Gbp/s (read %.2f map %.2f write %.2f s)" % (len(fs_), wall_f, fbases / wall_f / 1e9, st_f["t_read"], st_f["t_map"], st_f["t_write"]))
            except Exception as e:  # noqa: BLE001
                log("file-to-file leg error:", repr(e))
        if world == 1 and args.cpu_sample != 0:
            # one full step (the first timed batch) through the REAL reference on this host's cores: CPU baseline + parity
            bn, bs = batches[args.warmup]
            if args.cpu_sample > 0:
                bn, bs = bn[:args.cpu_sample], bs[:args.cpu_sample]
            try:
                out["cpu_baseline"], ref_paf = cpu_reference_baseline(fa, kf, bn, bs, cfg["preset"], tmp, n_cores)
                if ref_paf:
                    t2 = time.time()
                    ours, _, _, _ = mapper.map(bn, bs, copy_text=True)
                    with open(ref_paf, "rb") as f:
                        d = wmparity.diff_texts(f.read(), ours, sam=False, defined=wmparity.defined_names(bn, mapper.rep_len_defined()))
                    out["parity"] = {"reads": len(bs), "reads_with_hits": d["reads"], "hits": d["hits"], "mismatches": d["mismatches"], "mapq_compared": d["mapq_compared"],
                                     "compared": "PAF records incl. cg:Z CIGAR and all tags vs winnowmap_ref on the same reads; MAPQ and rl:i are compared wherever the reference assigns rep_len "
                                                 "(reads below 10 000 bases; above: the stage-2 rescan, src/map.c:808-813, and the fallback, :859-861) and masked on the pure two-stage "
                                                 "path, where its rep_len is an uninitialised stack word (src/map.c:281,933)"}
                    if d["examples"]:
                        out["parity"]["examples"] = d["examples"]
                    log("parity: %d reads, %d hits, %d mismatching reads (%.1fs)" % (len(bs), d["hits"], d["mismatches"], time.time() - t2))
            except Exception as e:  # noqa: BLE001
                log("cpu baseline / parity error:", repr(e))
                out.setdefault("cpu_baseline", None)
        print(json.dumps(out), flush=True)
    if os.environ.get("WM_PROF"):
        mapper.close()              # (prints the host glue's time per named region to stderr)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
