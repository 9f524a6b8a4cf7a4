#!/usr/bin/env python
"""Thread sweep of the REAL reference (oracle/_ref/winnowmap_ref) on one full bench step of a BASELINE config, on this host.
Writes gpurun_out/cpu_sweep_c<config>.json (+ a text table); the judged copies live under profiles/.
  python tools/cpu_sweep.py [--config 2] [--threads 16,32,64,128,256] [--reads N]"""
import argparse
import importlib.util
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)
from winnowmap_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--threads", default="")
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--ref-mb", type=float, default=250)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out"))
    a = ap.parse_args()
    cfg = B.CONFIGS[a.config]
    n_cores = os.cpu_count() or 1
    from winnowmap_amd import dist as wmdist
    quota = wmdist.available_cores()
    ths = [int(x) for x in a.threads.split(",") if x] or [t for t in (16, 32, 64, 96, 128, 192, 256, 384, 512) if t <= n_cores]
    tmp = tempfile.mkdtemp(prefix="wmsweep_")
    ref, fa, kf = B.make_workload(a.ref_mb, tmp)
    n = a.reads or cfg["reads_per_step"]
    t0 = time.time()
    reads, _ = synth.make_reads(ref, n, cfg["read_len"], cfg["seed"], profile=cfg["profile"], sv_frac=cfg["sv_frac"])
    seqs = [synth.codes_to_ascii(r) for r in reads]
    names = [b"r0_%d" % i for i in range(n)]
    rq = os.path.join(tmp, "step.fa")
    B.write_reads_fasta(rq, names, seqs)
    bases = sum(len(s) for s in seqs)
    B.log("%d reads written (%.1fs)" % (n, time.time() - t0))
    rows = []
    for sam in (False, True):
        for t in ths if not sam else sorted(rows, key=lambda r: r["map_s"])[:1] and [sorted(rows, key=lambda r: r["map_s"])[0]["threads"]]:
            r = B.run_reference(fa, kf, rq, cfg["preset"], t, os.path.join(tmp, "o.txt"), sam=sam)
            if r is None:
                continue
            rows.append({"threads": t, "format": "sam(-a)" if sam else "paf+cigar(-c)", "map_s": r[0], "index_s": r[1], "wall_s": r[2],
                         "gbps": bases / r[0] / 1e9, "reads_per_s": n / r[0]})
            B.log(json.dumps(rows[-1]))
    paf = [r for r in rows if r["format"].startswith("paf")]
    best = sorted(paf, key=lambda r: r["map_s"])
    out = {"config": a.config, "preset": cfg["preset"], "reads": n, "bases": bases, "cpu": B.cpu_model(), "host_threads": n_cores, "usable_cores": quota,
           "best_threads": [r["threads"] for r in best], "rows": rows}
    os.makedirs(a.out, exist_ok=True)
    with open(os.path.join(a.out, "cpu_sweep_c%d.json" % a.config), "w") as f:
        json.dump(out, f, indent=1)
    with open(os.path.join(a.out, "cpu_sweep_c%d.txt" % a.config), "w") as f:
        f.write("# winnowmap_ref thread sweep, config %d (%s), %d reads / %.3f Gbase, host: %s, %d hardware threads\n" % (a.config, cfg["preset"], n, bases / 1e9, out["cpu"], n_cores))
        f.write("# mapping phase = Real time - 'loaded/built the index' stamp (src/main.c:441,401)\n")
        f.write("%8s %-14s %10s %10s %10s %12s\n" % ("threads", "format", "map_s", "index_s", "Gbp/s", "reads/s"))
        for r in rows:
            f.write("%8d %-14s %10.2f %10.2f %10.4f %12.0f\n" % (r["threads"], r["format"], r["map_s"], r["index_s"], r["gbps"], r["reads_per_s"]))
    print(open(os.path.join(a.out, "cpu_sweep_c%d.txt" % a.config)).read())


if __name__ == "__main__":
    main()
