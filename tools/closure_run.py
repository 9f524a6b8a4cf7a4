#!/usr/bin/env python
"""BASELINE configs 4 and 5 at their reference sizes on ONE GPU: parity records, not throughput runs.

  python tools/closure_run.py config4 [--gb 3] [--reads 4096]     3-Gbase synthetic reference (24 contigs), -W list counted and index
                                                                 built on the device, 15-kb ONT-profile reads, PAF + CIGAR vs oracle/_ref/winnowmap_ref
  python tools/closure_run.py config5 [--contigs 24] [--ref-mb 600]   5-Mb contigs (asm20, k = 19; src/options.c:112-115) against a reference they
                                                                 were drawn from with 2 % divergence and structural variants

Each prints ONE JSON line (and appends it to the file given with --out): sizes, timings, what was compared, mismatches."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
from winnowmap_amd import gpu, parity, synth  # noqa: E402

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "winnowmap_ref")


def log(*a):
    print("[closure]", *a, file=sys.stderr, flush=True)


def write_reads(path, reads, prefix=b"r"):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b">" + prefix + b"%d\n" % i)
            f.write(synth.codes_to_ascii(r))
            f.write(b"\n")


def run_ref(args, out_path):
    t0 = time.time()
    with open(out_path, "wb") as fo:
        p = subprocess.run([REF_BIN] + args, stdout=fo, stderr=subprocess.PIPE)
    if p.returncode != 0:
        raise RuntimeError("winnowmap_ref failed: " + p.stderr.decode(errors="replace")[-600:])
    return time.time() - t0, p.stderr.decode(errors="replace")


def config4(args):
    tmp = tempfile.mkdtemp(prefix="wm_c4_")
    n_ctg = 24
    clen = int(args.gb * 1e9 / n_ctg)
    t0 = time.time()
    ref = synth.make_reference(n_ctg, clen, 71, repeat_frac=0.05)
    fa = os.path.join(tmp, "ref.fa")
    synth.write_fasta(fa, ref, prefix="chr")
    reads, _ = synth.make_reads(ref, args.reads, 15000, 72, profile="ont", sv_frac=0.01)
    # reads from the far end of the last contig (the largest offsets into the packed reference: beyond 2^31 bases)
    rng = np.random.default_rng(3)
    for j in range(8):
        tail = ref[n_ctg - 1][-(16000 + 1000 * j):-(1000 * j + 1)]
        reads[j] = synth.mutate_codes(tail[:15000].copy(), rng, 0.03, 0.03, 0.04)
    rq = os.path.join(tmp, "reads.fa")
    write_reads(rq, reads)
    seqs = [synth.codes_to_ascii(r) for r in reads]
    total = sum(len(c) for c in ref)
    del ref
    t_gen = time.time() - t0
    log("reference %d x %d = %.2f Gbase, %d reads (%.0f s)" % (n_ctg, clen, total / 1e9, len(seqs), t_gen))
    ctx = gpu.Context(0, int(args.arena_gb) << 30)
    kf = os.path.join(tmp, "rep.txt")
    t0 = time.time()
    n_k, st = gpu.write_repetitive_kmers_gpu(ctx, fa, 15, kf)
    t_w = time.time() - t0
    log("-W list on the device: %d k-mers (%.1f s)" % (n_k, t_w))
    t0 = time.time()
    idx, ist = gpu.Index.build_on_device(ctx, fa, kf, k=15, w=50, n_threads=16)
    t_idx = time.time() - t0
    log("index on the device: %d minimizers, %.1f s (%s)" % (idx.n_minimizers, t_idx, ist))
    idx.upload(ctx)
    m = gpu.Mapper(ctx, idx, "map-ont", gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG)
    m.set_threads(16, 8 << 30)
    t0 = time.time()
    ours, hits, _, _ = m.map([b"r%d" % i for i in range(len(seqs))], seqs)
    t_map = time.time() - t0
    log("mapped %d reads, %d hits (%.1f s)" % (len(seqs), len(hits), t_map))
    ref_paf = os.path.join(tmp, "ref.paf")
    t_ref, err = run_ref(["-t", "16", "-W", kf, "-cx", "map-ont", fa, rq], ref_paf)
    log("reference binary: %.0f s" % t_ref)
    with open(ref_paf, "rb") as f:
        d = parity.diff_texts(f.read(), ours, sam=False)
    far = sum(1 for l in ours.split(b"\n") if l.startswith(b"r") and l.split(b"\t")[0] in (b"r%d" % j for j in range(8)) and l.split(b"\t")[5] == b"chr%d" % (n_ctg - 1))
    n_mini = int(idx.n_minimizers)
    m.close(); idx.close(); ctx.close()
    return {"record": "BASELINE config 4 reference at size, one MI355X", "reference_gbase": total / 1e9, "contigs": n_ctg, "reads": len(seqs), "read_len": 15000,
            "kmer_list": {"where": "device", "k": 15, "kmers": int(n_k), "seconds": round(t_w, 1)},
            "index": {"where": "device (sketch + table)", "minimizers": n_mini, "seconds": round(t_idx, 1), "stats": ist},
            "map_seconds": round(t_map, 1), "reference_binary_seconds": round(t_ref, 1), "records_of_reads_from_beyond_2^31_bases": far,
            "parity": {"reads_with_hits": d["reads"], "hits": d["hits"], "mismatches": d["mismatches"], "examples": d.get("examples", [])[:3],
                       "compared": "PAF incl. cg:Z vs winnowmap_ref -t 16 -W -cx map-ont on the same files (MAPQ / rl:i masked for reads >= 10 kb, parity.py)"}}


def config5(args):
    tmp = tempfile.mkdtemp(prefix="wm_c5_")
    n_ctg_ref = max(1, int(round(args.ref_mb / 50.0)))
    t0 = time.time()
    ref = synth.make_reference(n_ctg_ref, int(args.ref_mb * 1e6 / n_ctg_ref), 81, repeat_frac=0.05)
    fa = os.path.join(tmp, "ref.fa")
    synth.write_fasta(fa, ref, prefix="chr")
    rng = np.random.default_rng(82)
    clen = args.contig_mb * 1000000
    reads = []
    for i in range(args.contigs):            # as tests/test_binding_gpu.py::test_parity_at_scale_config5_shape_asm20, five times longer: 5 % divergence, 1 SV per 100 kb
        c = ref[int(rng.integers(0, len(ref)))]
        st = int(rng.integers(0, len(c) - clen))
        q = synth.mutate_codes(c[st:st + clen].copy(), rng, 0.03, 0.01, 0.01)
        for _ in range(clen // 100000):
            p = int(rng.integers(10000, len(q) - 10000))
            q = np.concatenate([q[:p], q[p + 2000:]]) if rng.integers(0, 2) else np.concatenate([q[:p], synth.random_codes(1000, rng), q[p:]])
        reads.append(q if i % 2 else synth.revcomp_codes(q))
    rq = os.path.join(tmp, "contigs.fa")
    write_reads(rq, reads, b"ctg")
    seqs = [synth.codes_to_ascii(r) for r in reads]
    total = sum(len(c) for c in ref)
    del ref
    log("reference %.0f Mb, %d contigs of %d Mb (%.0f s)" % (total / 1e6, len(seqs), args.contig_mb, time.time() - t0))
    ref_paf = os.path.join(tmp, "ref.paf")
    t_ref, t_ref_map, err = 0.0, None, ""
    if not args.skip_ref:
        t_ref, err = run_ref(["-t", "16", "-cx", "asm20", fa, rq], ref_paf)
        import re
        m_idx = re.search(r"\[M::main::([0-9.]+)\*[0-9.]+\] loaded/built the index", err)
        m_end = re.search(r"Real time: ([0-9.]+) sec", err)
        if m_idx and m_end:
            t_ref_map = float(m_end.group(1)) - float(m_idx.group(1))          # the reference's mapping phase alone (src/main.c:401, :441)
        log("reference binary: %.0f s, of which mapping %s s" % (t_ref, "%.1f" % t_ref_map if t_ref_map is not None else "?"))
    ctx = gpu.Context(0, int(args.arena_gb) << 30)
    t0 = time.time()
    idx, ist = gpu.Index.build_on_device(ctx, fa, None, k=19, w=50, n_threads=16)
    t_idx = time.time() - t0
    log("index on the device: %d minimizers, %.1f s" % (idx.n_minimizers, t_idx))
    idx.upload(ctx)
    m = gpu.Mapper(ctx, idx, "asm20", gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG)
    m.set_threads(16, 16 << 30)
    t0 = time.time()
    ours, hits, _, _ = m.map([b"ctg%d" % i for i in range(len(seqs))], seqs)
    t_map = time.time() - t0
    log("mapped %d contigs, %d hits (%.1f s)" % (len(seqs), len(hits), t_map))
    if args.skip_ref:
        d = {"reads": 0, "hits": len(hits), "mismatches": 0, "examples": []}
    else:
        with open(ref_paf, "rb") as f:
            d = parity.diff_texts(f.read(), ours, sam=False)
    ks = m.kernel_stats()
    hs = m.host_stats()
    n_mini = int(idx.n_minimizers)
    m.close(); idx.close(); ctx.close()
    wide = {k: v for k, v in ks.items() if v[2] > 0}
    return {"record": "BASELINE config 5 contig size, one MI355X", "reference_mb": total / 1e6, "contigs": len(seqs), "contig_mb": args.contig_mb, "preset": "asm20 (k 19, w 50: src/options.c:112-115 leaves w at its default)",
            "index": {"where": "device", "minimizers": n_mini, "seconds": round(t_idx, 1)}, "map_seconds": round(t_map, 1), "gbps": sum(len(s) for s in seqs) / t_map / 1e9,
            "reference_binary_seconds": round(t_ref, 1), "reference_mapping_seconds": None if t_ref_map is None else round(t_ref_map, 1), "host": hs,
            "ksw_classes_used": {str(k): {"ms": round(v[0], 1), "cells": v[1], "launches": v[2]} for k, v in wide.items()},
            "parity": {"reads_with_hits": d["reads"], "hits": d["hits"], "mismatches": d["mismatches"], "examples": d.get("examples", [])[:3],
                       "compared": "PAF incl. cg:Z vs winnowmap_ref -t 16 -cx asm20 on the same files"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["config4", "config5"])
    ap.add_argument("--gb", type=float, default=3.0)
    ap.add_argument("--reads", type=int, default=4096)
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--contig-mb", type=int, default=5)
    ap.add_argument("--ref-mb", type=float, default=600.0)
    ap.add_argument("--arena-gb", type=float, default=40.0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--skip-ref", action="store_true", help="config5: no reference run, no parity (profiling runs)")
    args = ap.parse_args()
    if not os.path.exists(REF_BIN):
        sys.exit("oracle/_ref/winnowmap_ref is not built (python -m winnowmap_amd.build in the container that has /root/reference)")
    t0 = time.time()
    rec = config4(args) if args.what == "config4" else config5(args)
    rec["wall_seconds"] = round(time.time() - t0, 1)
    line = json.dumps(rec)
    print(line, flush=True)
    if args.out:
        with open(args.out, "a") as f:
            f.write(line + "\n")
    sys.exit(0 if rec["parity"]["mismatches"] == 0 and rec["parity"]["hits"] > 0 else 3)


if __name__ == "__main__":
    main()
