"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/wm_gpu.h declares, and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re
import numpy as np
import pytest
import wmtest as W
from winnowmap_amd import build, gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_gpu()
    return gpu.lib()


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "wm_gpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wm_[a-z0-9_]+)\s*\(", txt)))


def test_exports_every_declared_symbol(lib):
    syms = declared_symbols()
    assert len(syms) >= 10
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = lib.wm_ctx_create(0, 0, C.byref(h))
    assert rc == -1 and not h.value            # WM_ENODEV
    assert b"no HIP device" in lib.wm_last_error()
    with pytest.raises(gpu.WmError):
        gpu.Context(0)


def test_product_does_not_reference_oracle():
    # the product tree must never include / link / import anything under oracle/ or tests/
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "winnowmap_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                if f == "build.py":      # builds the checkers for the tests, never loads them
                    continue
                for line in open(os.path.join(d, f), errors="ignore"):
                    ls = line.strip()
                    if not (ls.startswith(("#include", "import ", "from ")) or "CDLL(" in ls or "dlopen(" in ls):
                        continue
                    for needle in ("oracle", "libwinnowmap_ref", "simt_emu", "wmtest"):
                        if needle in ls:
                            bad.append((f, ls))
    assert not bad, bad


def test_presets_equal_the_references_mm_set_opt():
    """wm_mapopt_preset (host code of the library, no GPU needed) against mm_set_opt of the real reference (oracle/_ref): every mirrored
    mm_mapopt_t field, and k / w, for every preset the library accepts."""
    import ctypes as C
    import numpy as np
    import pytest
    import wmtest as W
    if not W.have_ref():
        pytest.skip("oracle/_ref not built")
    R = W.ref()
    R.refshim_preset_fields.argtypes = [C.c_char_p, C.c_void_p, C.c_int]
    build.build_gpu()
    L = C.CDLL(gpu.LIB_PATH)

    class MapOpt(C.Structure):
        _fields_ = [("flag", C.c_int64)] + [(n, C.c_int32) for n in ("seed", "sdust_thres", "max_qlen", "bw", "max_gap", "max_gap_ref", "min_gap_ref", "max_frag_len",
                                                                      "max_chain_skip", "max_chain_iter", "min_cnt", "min_chain_score")] + \
                   [("chain_gap_scale", C.c_float)] + [(n, C.c_int32) for n in ("SVaware", "SVawareMinReadLength", "suffixSampleOffset", "min_mapq")] + [("min_qcov", C.c_float)] + \
                   [(n, C.c_int32) for n in ("minPrefixLength", "maxPrefixLength")] + [("prefixIncrementFactor", C.c_float)] + \
                   [(n, C.c_int32) for n in ("stage2_bw", "stage2_zdrop_inv", "stage2_max_gap")] + [("mask_level", C.c_float), ("mask_len", C.c_int32), ("pri_ratio", C.c_float), ("best_n", C.c_int32)] + \
                   [(n, C.c_int32) for n in ("max_join_long", "max_join_short", "min_join_flank_sc")] + [("min_join_flank_ratio", C.c_float), ("alt_drop", C.c_float)] + \
                   [(n, C.c_int32) for n in ("a", "b", "q", "e", "q2", "e2", "sc_ambi", "zdrop", "zdrop_inv", "end_bonus", "min_dp_max", "min_ksw_len")] + \
                   [("max_clip_ratio", C.c_float), ("mid_occ_frac", C.c_float)] + [(n, C.c_int32) for n in ("min_mid_occ", "mid_occ", "max_occ")] + [("mini_batch_size", C.c_int64), ("max_sw_mat", C.c_int64)] + \
                   [(n, C.c_int32) for n in ("noncan", "junc_bonus", "anchor_ext_len", "anchor_ext_shift")]
    L.wm_mapopt_preset.argtypes = [C.c_char_p, C.POINTER(MapOpt), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    for preset in (b"", b"map-ont", b"map-pb", b"map-pb-clr", b"asm5", b"asm10", b"asm20", b"splice", b"splice:hq", b"cdna"):
        o = MapOpt(); k = C.c_int(); w = C.c_int()
        assert L.wm_mapopt_preset(preset, C.byref(o), C.byref(k), C.byref(w)) == 0
        ours = [float(getattr(o, n)) for n, _ in MapOpt._fields_] + [float(k.value), float(w.value)]
        ref = np.zeros(64)
        n = R.refshim_preset_fields(preset, ref.ctypes.data, 64)
        assert n == len(ours), (n, len(ours))
        for (name, _), a, b_ in zip(list(MapOpt._fields_) + [("k", 0), ("w", 0)], ours, ref[:n]):
            assert a == np.float32(b_) or a == b_, (preset, name, a, b_)
    assert L.wm_mapopt_preset(b"no-such-preset", C.byref(MapOpt()), None, None) != 0


@pytest.mark.skipif(not W.have_ref(), reason="oracle/_ref not built")
def test_exported_ksw_ll_i16_equals_the_references():
    """wm_ksw_ll_i16 (include/wm_gpu.h; host code, callable without a GPU) against ksw_ll_qinit + ksw_ll_i16 of the reference (src/ksw2.h:82-83):
    score and both end coordinates, incl. the striped layout's negative query ends."""
    build.build_gpu()
    L = C.CDLL(gpu.LIB_PATH)
    L.wm_ksw_ll_i16.restype = C.c_int
    L.wm_ksw_ll_i16.argtypes = [C.c_int, W.u8p, C.c_int, W.u8p, W.i8p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    from winnowmap_amd import synth
    rng = np.random.default_rng(12)
    for it in range(300):
        q = rng.integers(0, 5 if it % 7 == 0 else 4, int(rng.integers(1, 500))).astype(np.uint8)
        t = rng.integers(0, 4, int(rng.integers(1, 500))).astype(np.uint8) if it % 2 == 0 else synth.mutate_codes(q, rng, 0.05, 0.05, 0.05)
        if len(t) == 0:
            t = q
        a, b = (1, 4) if it % 3 else (2, 4)
        mat = W.simple_mat(a, b, 1)
        go, ge = (4, 2) if it % 3 else (6, 1)
        qe, te = C.c_int(), C.c_int()
        s = L.wm_ksw_ll_i16(len(q), q, len(t), t, mat, go, ge, C.byref(qe), C.byref(te))
        assert (s, qe.value, te.value) == W.r_ksw_ll(q, t, mat, go, ge), it


def test_round4_entry_points_refuse_bad_arguments_without_touching_a_device(lib):
    """argument checks of the entry points added in round 4 run before any device call: they must answer WM_EINVAL (-2) with a message, on any box"""
    lib.wm_map_file_multi.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int64, C.c_void_p]
    assert lib.wm_map_file_multi(None, 0, b"reads.fa", b"-", 0, None) == -2 and lib.wm_last_error()
    lib.wm_debug_stripe_timing.argtypes = [C.c_void_p, C.c_int]
    out = np.zeros(16, np.uint64)
    assert lib.wm_debug_stripe_timing(out.ctypes.data, 0) == -2          # the shipped library is not the WM_STRIPE_TIMING variant
    assert b"WM_STRIPE_TIMING" in lib.wm_last_error()
    lib.wm_ksw_batch_pos_zd.argtypes = [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 3 + [C.c_size_t, C.c_void_p, C.c_void_p]
    assert lib.wm_ksw_batch_pos_zd(None, None, 3, None, None, None, 0, None, None) == -2
    lib.wm_ksw_set_routing(-1, -1, -1)                                    # (leaves everything as it is; must not need a device)
    lib.wm_set_cmdline.argtypes = [C.c_int, C.c_void_p]
    lib.wm_set_cmdline(-1, None)
    lib.wm_set_cmdline(0, None)
