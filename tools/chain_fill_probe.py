"""On-GPU timing of the dense chain fill (wm_chain_batch) on satellite-like anchor sets: us per anchor for the round-5 workgroup kernel (WM_CHAIN_WIDE=0) and the
whole-window kernel in several geometries (WM_CHAIN_WIDE_GEOM). Each variant runs in a process of its own (the switches are read once).
   python tools/chain_fill_probe.py            the sweep
   python tools/chain_fill_probe.py one        one measurement with the environment as it is"""
import sys, os, time, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def anchors(seed, n_mini, copies, period=171):
    rng = np.random.default_rng(seed)
    qpos = np.cumsum(rng.integers(5, 40, n_mini)).astype(np.int64) + 100
    k = rng.integers(-copies, copies + 1, (n_mini, copies)).astype(np.int64)
    x = (1_000_000 + qpos[:, None] + k * period + rng.integers(-2, 3, (n_mini, copies))).ravel().astype(np.uint64)
    y = (np.uint64(15 << 32) | np.repeat(qpos, copies).astype(np.uint64))
    o = np.argsort(x, kind="stable")
    return x[o], y[o]


def one():
    from winnowmap_amd import gpu
    import wmtest as W
    M128 = np.dtype([("x", np.uint64), ("y", np.uint64)])
    PAR = np.dtype([("p", np.int32, 8), ("gs", np.float32), ("is_cdna", np.int32)])      # wm_chain_par_t (include/wm_gpu.h)
    ctx = gpu.Context(0, 8 << 30)
    L = gpu.lib()
    L.wm_chain_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, W.u64p, W.i32p, C.c_void_p, W.u64p, W.u64p, W.i32p, W.i32p]
    L.wm_last_aux_ms.restype = C.c_float; L.wm_last_aux_ms.argtypes = [C.c_void_p]
    for name, n_mini, copies in (("dense 5000-window", 4000, 64), ("medium 600-window", 20000, 8)):
        x, y = anchors(7, n_mini, copies)
        a = np.zeros(len(x), M128); a["x"], a["y"] = x, y
        par = np.zeros(1, PAR); par["p"][0] = (16000, 1000, 16000, 2000, 25, 5000, 3, 40); par["gs"][0] = 1.0
        aoff = np.zeros(1, np.uint64); na = np.array([len(a)], np.int32)
        u = np.zeros(len(a) + 1, np.uint64); uoff = np.zeros(1, np.uint64); nu = np.zeros(1, np.int32); nv = np.zeros(1, np.int32)
        best = 1e9
        for _ in range(2):
            a2 = a.copy()
            assert L.wm_chain_batch(ctx._h, 1, a2.ctypes.data, aoff, na, par.ctypes.data, u, uoff, nu, nv) == 0, L.wm_last_error()
            best = min(best, L.wm_last_aux_ms(ctx._h))
        print("%-22s n=%d  kernel %.1f ms = %.2f us per anchor  (chains %d, anchors kept %d, checksum %d)" % (name, len(a), best, best * 1e3 / len(a), nu[0], nv[0], int(u[:nu[0]].sum() % 1000003)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        for env in ({"WM_CHAIN_WIDE": "0"}, {"WM_CHAIN_WIDE_GEOM": "16x5"}, {"WM_CHAIN_WIDE_GEOM": "16x5", "WM_CHAIN_WIDE_FIRST": "2"}, {"WM_CHAIN_WIDE_GEOM": "16x5", "WM_CHAIN_WIDE_FIRST": "5"},
                    {"WM_CHAIN_WIDE_GEOM": "8x10"}, {"WM_CHAIN_WIDE_GEOM": "16x3"}, {"WM_CHAIN_WIDE_GEOM": "8x5"}, {"WM_CHAIN_WIDE_GEOM": "4x10"}):
            print("==", env, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, **env))
