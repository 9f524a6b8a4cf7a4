"""On-GPU micro-benchmark of the chain kernel on anchor sets from repeat-rich reads (satellite arrays)."""
import sys, os, time, tempfile, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from winnowmap_amd import gpu, synth
import wmtest as W

M128 = np.dtype([("x", np.uint64), ("y", np.uint64)])
tmp = tempfile.mkdtemp()
ref = synth.make_reference(1, 3_000_000, 3, repeat_frac=0.5)       # half of it repeats: long satellite arrays
synth.write_fasta(tmp + "/ref.fa", ref)
use_w = len(sys.argv) > 1 and sys.argv[1] == "W"
kf = None
if use_w:
    km, cnt = synth.repetitive_kmers(ref, 15)
    kf = tmp + "/rep.txt"; synth.write_kmer_list(kf, km, cnt, 15)
ctx = gpu.Context(0, 8 << 30)
idx = gpu.Index(tmp + "/ref.fa", kf, k=15, w=50); idx.upload(ctx)
L = gpu.lib()
L.wm_sketch_batch.argtypes = [C.c_void_p, C.c_int, W.u8p, C.c_size_t, W.u64p, W.i32p, C.c_void_p, C.c_size_t, W.u64p, W.i32p]
L.wm_seed_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, W.u64p, W.i32p, W.i32p, C.c_int, C.c_int64, C.c_void_p, C.c_size_t, W.u64p, W.i32p, W.i32p]
L.wm_chain_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, W.u64p, W.i32p, C.c_void_p, W.u64p, W.u64p, W.i32p, W.i32p]
reads, _ = synth.make_reads(ref, 64, 15000, 5)
seqs = [r[st:st + 2000].copy() for r in reads for st in range(0, 14000, 2000)] + list(reads[:16])
n = len(seqs)
lens = np.array([len(s) for s in seqs], np.int32); offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
allseq = np.concatenate(seqs)
out = np.zeros(int(lens.sum()) + n, M128); ooff = np.zeros(n, np.uint64); cnt = np.zeros(n, np.int32)
assert L.wm_sketch_batch(ctx._h, n, allseq, allseq.nbytes, offs, lens, out.ctypes.data, len(out), ooff, cnt) == 0
nm = cnt.copy(); moff = ooff.copy()
cap = 80_000_000
aout = np.zeros(cap, M128); aoff = np.zeros(n, np.uint64); na = np.zeros(n, np.int32); rl = np.zeros(n, np.int32)
assert L.wm_seed_batch(ctx._h, n, out.ctypes.data, moff, nm, lens, 5000, 0, aout.ctypes.data, cap, aoff, na, rl) == 0, L.wm_last_error()
print("anchor counts: median %d p90 %d max %d total %d (seed kernel %.1f ms)" % (np.median(na), np.percentile(na, 90), na.max(), na.sum(), L.wm_last_aux_ms(ctx._h)))
PAR = np.dtype([("p", np.int32, 8), ("gs", np.float32), ("is_cdna", np.int32)])      # wm_chain_par_t (include/wm_gpu.h)
for sel_name, sel in (("all", np.arange(n)), ("largest", np.argsort(-na)[:1]), ("n<=256", np.nonzero(na <= 256)[0]), ("1024<n<=4096", np.nonzero((na > 1024) & (na <= 4096))[0]), ("n>4096", np.nonzero(na > 4096)[0])):
    if len(sel) == 0: continue
    parts = [aout[int(aoff[i]):int(aoff[i]) + na[i]] for i in sel]
    na2 = np.array([len(a) for a in parts], np.int32); aoff2 = np.concatenate([[0], np.cumsum(na2)[:-1]]).astype(np.uint64)
    alla = np.concatenate(parts)
    par = np.zeros(len(sel), PAR); par["p"] = [5000, 1000, 5000, 500, 25, 5000, 3, 40]; par["gs"] = 1.0
    u = np.zeros(len(alla) + 1, np.uint64); uoff = np.zeros(len(sel), np.uint64); nu = np.zeros(len(sel), np.int32); nv = np.zeros(len(sel), np.int32)
    best = 1e9
    for rep in range(2):
        a2 = alla.copy()
        assert L.wm_chain_batch(ctx._h, len(sel), a2.ctypes.data, aoff2, na2, par.ctypes.data, u, uoff, nu, nv) == 0, L.wm_last_error()
        best = min(best, L.wm_last_aux_ms(ctx._h))
    print("%-14s jobs=%d anchors=%d chain kernel %.2f ms -> %.2f us/anchor" % (sel_name, len(sel), na2.sum(), best, best * 1e3 / max(1, na2.sum())), flush=True)
i = int(np.argmax(na)); a = aout[int(aoff[i]):int(aoff[i]) + na[i]]
t0 = time.time(); W.o_chain_dp(a["x"], a["y"]); print("oracle (1 core) largest job n=%d: %.1f ms" % (na[i], (time.time() - t0) * 1e3))
