#!/usr/bin/env python
"""One contig that crosses a tandem satellite array (171-bp monomers at 2 % per-copy divergence) mapped with asm20 against the reference it was drawn from:
the window stage's worst case (10^5 .. 10^6 anchors in one job, dense). Prints the mapping time and checks the records against the reference binary.
  python tools/satellite_probe.py [--array-mb 1.0] [--flank-kb 300]
Meant to run under rocprofv3 --kernel-trace --stats (tools/r05h_run.sh) to see which kernel owns the call."""
import argparse, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
from winnowmap_amd import gpu, parity, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--array-mb", type=float, default=1.0)
ap.add_argument("--flank-kb", type=int, default=300)
ap.add_argument("--preset", default="asm20")
args = ap.parse_args()
rng = np.random.default_rng(5)
tmp = tempfile.mkdtemp(prefix="wm_sat_")
ref = [synth.random_codes(40_000_000, rng) for _ in range(2)]
mono = synth.random_codes(171, rng)
ncopy = int(args.array_mb * 1e6) // 171
arr = np.concatenate([synth.mutate_codes(mono, rng, 0.02, 0.0, 0.0) for _ in range(ncopy)])
p0 = 10_000_000
ref[0][p0:p0 + len(arr)] = arr
fa = os.path.join(tmp, "ref.fa")
synth.write_fasta(fa, ref, prefix="chr")
fl = args.flank_kb * 1000
q = synth.mutate_codes(ref[0][p0 - fl:p0 + len(arr) + fl].copy(), rng, 0.03, 0.01, 0.01)
k = 19 if args.preset.startswith("asm") else 15
rq = os.path.join(tmp, "q.fa")
with open(rq, "wb") as f:
    f.write(b">ctg0\n" + synth.codes_to_ascii(q) + b"\n")
t0 = time.time()
refbin = os.path.join(ROOT, "oracle", "_ref", "winnowmap_ref")
p = subprocess.run([refbin, "-t", "16", "-cx", args.preset, fa, rq], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
t_ref = time.time() - t0
ctx = gpu.Context(0, 40 << 30)
idx, ist = gpu.Index.build_on_device(ctx, fa, None, k=k, w=50, n_threads=16)
idx.upload(ctx)
m = gpu.Mapper(ctx, idx, args.preset, gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG)
m.set_threads(16, 16 << 30)
seq = synth.codes_to_ascii(q)
m.map([b"ctg0"], [seq])            # warm-up (allocations)
t0 = time.time()
ours, hits, _, _ = m.map([b"ctg0"], [seq])
t_map = time.time() - t0
d = parity.diff_texts(p.stdout, ours, sam=False)
print("contig %d bp across a %.2f-Mb satellite array: mapped in %.2f s (%d hits), reference binary %.1f s wall incl. its index, parity mismatches %d" % (len(q), len(arr) / 1e6, t_map, len(hits), t_ref, d["mismatches"]), flush=True)
m.close(); idx.close(); ctx.close()
