"""GPU: the dense chain fill (wm_chain_batch) in every geometry against the oracle's mm_chain_dp on satellite-like anchor sets of growing size.
   python tools/chain_fill_check.py            (each geometry runs in a process of its own: the switch is read once)"""
import sys, os, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
SIZES = ((250, 64), (1000, 64), (4000, 64), (4000, 16), (20000, 8))


def one():
    from winnowmap_amd import gpu
    import wmtest as W
    import chain_fill_probe as P
    M128 = np.dtype([("x", np.uint64), ("y", np.uint64)])
    PAR = np.dtype([("p", np.int32, 8), ("gs", np.float32), ("is_cdna", np.int32)])      # wm_chain_par_t (include/wm_gpu.h)
    ctx = gpu.Context(0, 8 << 30)
    L = gpu.lib()
    L.wm_chain_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, W.u64p, W.i32p, C.c_void_p, W.u64p, W.u64p, W.i32p, W.i32p]
    out = []
    for n_mini, copies in SIZES:
        x, y = P.anchors(7, n_mini, copies)
        ou, obx, oby = W.o_chain_dp(x, y, max_dist_x=16000, min_dist_x=1000, max_dist_y=16000, bw=2000, max_skip=25, max_iter=5000)
        a = np.zeros(len(x), M128); a["x"], a["y"] = x, y
        par = np.zeros(1, PAR); par["p"][0] = (16000, 1000, 16000, 2000, 25, 5000, 3, 40); par["gs"][0] = 1.0
        aoff = np.zeros(1, np.uint64); na = np.array([len(a)], np.int32)
        u = np.zeros(len(a) + 1, np.uint64); uoff = np.zeros(1, np.uint64); nu = np.zeros(1, np.int32); nv = np.zeros(1, np.int32)
        assert L.wm_chain_batch(ctx._h, 1, a.ctypes.data, aoff, na, par.ctypes.data, u, uoff, nu, nv) == 0, L.wm_last_error()
        ok = nu[0] == len(ou) and np.array_equal(u[:nu[0]], ou) and np.array_equal(a["x"][:nv[0]], obx)
        first = -1
        if not ok:
            m = min(nu[0], len(ou))
            d = np.nonzero(u[:m] != ou[:m])[0]
            first = int(d[0]) if len(d) else m
        out.append("n=%d %s%s" % (len(x), "ok" if ok else "MISMATCH", "" if ok else " (chains %d vs %d, first differing chain %d)" % (nu[0], len(ou), first)))
    print("  ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        geoms = sys.argv[1:] or ["0", "16x5", "8x5", "4x10", "16x3", "8x10", "4x5", "2x10", "1x10"]
        for g in geoms:
            env = {"WM_CHAIN_WIDE": "0"} if g == "0" else {"WM_CHAIN_WIDE_GEOM": g}
            print("==", env, end="  ", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, **env))
