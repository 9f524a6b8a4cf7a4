"""On-GPU timing of wm_chain_batch on chain jobs dumped from the host harness (WM_CHAIN_DUMP; bench-like workload).
usage: chain_probe2.py bench_data/chain_jobs.bin [replicas]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from winnowmap_amd import gpu
import wmtest as W

M128 = np.dtype([("x", np.uint64), ("y", np.uint64)])
PAR = np.dtype([("p", np.int32, 8), ("gs", np.float32), ("is_cdna", np.int32)])      # wm_chain_par_t (include/wm_gpu.h)
buf = open(sys.argv[1], "rb").read()
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 1
jobs = []
o = 0
while o < len(buf):
    n = int(np.frombuffer(buf, np.int64, 1, o)[0]); o += 16
    par = np.frombuffer(buf, np.int32, 10, o).copy(); o += 40
    a = np.frombuffer(buf, M128, n, o).copy(); o += 16 * n
    jobs.append((n, par, a))
jobs = jobs * rep
na = np.array([j[0] for j in jobs], np.int32)
print("jobs %d anchors %d  n>4096: %d jobs %d anchors; 1024<n<=4096: %d jobs" % (len(jobs), na.sum(), (na > 4096).sum(), na[na > 4096].sum(), ((na > 1024) & (na <= 4096)).sum()))
ctx = gpu.Context(0, 8 << 30)
L = gpu.lib()
L.wm_chain_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, W.u64p, W.i32p, C.c_void_p, W.u64p, W.u64p, W.i32p, W.i32p]


def run(sel, label, check=False):
    parts = [jobs[i][2] for i in sel]
    na2 = np.array([len(a) for a in parts], np.int32); aoff2 = np.concatenate([[0], np.cumsum(na2)[:-1]]).astype(np.uint64)
    alla = np.concatenate(parts)
    par = np.zeros(len(sel), PAR)
    for k, i in enumerate(sel):
        par["p"][k] = jobs[i][1][:8]; par["gs"][k] = jobs[i][1][8:9].view(np.float32)[0]
    u = np.zeros(len(alla) + 1, np.uint64); uoff = np.zeros(len(sel), np.uint64); nu = np.zeros(len(sel), np.int32); nv = np.zeros(len(sel), np.int32)
    best = 1e9; wall = 1e9
    for _ in range(2):
        a2 = alla.copy()
        t0 = time.time()
        assert L.wm_chain_batch(ctx._h, len(sel), a2.ctypes.data, aoff2, na2, par.ctypes.data, u, uoff, nu, nv) == 0, L.wm_last_error()
        wall = min(wall, (time.time() - t0) * 1e3)
        best = min(best, L.wm_last_aux_ms(ctx._h))
    if check:
        for k, i in enumerate(sel[:6]):
            a = jobs[i][2]; pp = jobs[i][1]
            uo, bx, by = W.o_chain_dp(a["x"], a["y"], *[int(x) for x in pp[:8]], gap_scale=float(pp[8:9].view(np.float32)[0]))
            assert nu[k] == len(uo) and nv[k] == len(bx), ("count mismatch", k, nu[k], len(uo), nv[k], len(bx))
            assert np.array_equal(u[int(uoff[k]):int(uoff[k]) + nu[k]], uo) and np.array_equal(a2["x"][int(aoff2[k]):int(aoff2[k]) + nv[k]], bx)
        print("   (first jobs match the oracle)")
    print("%-34s jobs=%d anchors=%d kernel %.2f ms (%.3f us/anchor), call wall %.1f ms" % (label, len(sel), na2.sum(), best, best * 1e3 / max(1, na2.sum()), wall), flush=True)
    return nu.copy(), nv.copy(), u.copy()


big = np.nonzero(na > 4096)[0]
one = big[np.argsort(-na[big])[:1]]
run(big, "n>4096", check=True)
run(one, "  largest (n=%d)" % na[one[0]])
run(np.arange(len(jobs)), "all jobs")
run(np.nonzero(na <= 256)[0], "n<=256")
t0 = time.time(); a = jobs[one[0]][2]; W.o_chain_dp(a["x"], a["y"]); print("oracle (1 core) largest job n=%d: %.1f ms" % (na[one[0]], (time.time() - t0) * 1e3))
