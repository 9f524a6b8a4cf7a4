"""On-GPU probe of the whole mapper: wall vs kernel time for a batch of synthetic ONT reads."""
import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from winnowmap_amd import gpu, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ref_mb = float(sys.argv[2]) if len(sys.argv) > 2 else 20
t0 = time.time()
tmp = tempfile.mkdtemp()
ref = synth.make_reference(max(1, int(ref_mb / 10)), 10_000_000 if ref_mb >= 10 else int(ref_mb * 1e6), 3, repeat_frac=0.1)
synth.write_fasta(tmp + "/ref.fa", ref)
t1 = time.time()
ctx = gpu.Context(0, 48 << 30)
kf = None
if len(sys.argv) > 3 and sys.argv[3] == "W":
    tk = time.time()
    km, cnt = synth.repetitive_kmers(ref, 15)
    kf = tmp + "/rep.txt"
    synth.write_kmer_list(kf, km, cnt, 15)
    print("-W list: %d k-mers (%.1fs)" % (len(km), time.time() - tk), flush=True)
idx = gpu.Index(tmp + "/ref.fa", kf, k=15, w=50, n_threads=32)
t2 = time.time()
idx.upload(ctx)
reads, _ = synth.make_reads(ref, n, 15000, 7, sv_frac=0.01)
seqs = [synth.codes_to_ascii(r) for r in reads]
names = ["r%d" % i for i in range(n)]
t3 = time.time()
print("gen ref %.1fs, index %.1fs (%d minimizers), reads %.1fs" % (t1 - t0, t2 - t1, idx.n_minimizers, t3 - t2), flush=True)
m = gpu.Mapper(ctx, idx, "map-ont", gpu.MM_F_CIGAR | gpu.MM_F_OUT_CG)
nth = int(sys.argv[4]) if len(sys.argv) > 4 else 1
if nth > 1:
    m.set_threads(nth, 48 << 30)
for rep in range(2):
    t4 = time.time()
    text, hits, cig, first = m.map(names, seqs)
    t5 = time.time()
    st = m.stats()
    print("batch of %d reads: wall %.3f s -> %.4f Gbp/s | kernels: ksw %.3f s, aux %.3f s | cells %.3e (%.1f GCUPS on ksw time) | super-steps %d, ksw jobs %d, hits %d" %
          (n, t5 - t4, st["read_bases"] / (t5 - t4) / 1e9, st["ksw_kernel_us"] / 1e6, st["aux_kernel_us"] / 1e6, st["dp_cells"],
           st["dp_cells"] / max(st["ksw_kernel_us"], 1) / 1e3, st["super_steps"], st["ksw_jobs"], len(hits)), flush=True)
