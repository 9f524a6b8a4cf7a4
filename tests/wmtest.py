"""Test-side ctypes bindings for the checkers: oracle/libwm_oracle.so (our CPU restatement) and, when it
has been built (oracle/_ref/, build container or prebuilt on the GPU box), the REAL reference library.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libwm_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libwinnowmap_ref.so")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "winnowmap_ref")

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")

EZ_FIELDS = ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score", "reach_end")


def simple_mat(a=2, b=4, sc_ambi=1):
    """5x5 match/mismatch/N matrix as src/align.c:9 builds it."""
    m = np.full((5, 5), -abs(b), np.int8)
    for i in range(4):
        m[i, i] = abs(a)
    m[4, :] = -abs(sc_ambi)
    m[:, 4] = -abs(sc_ambi)
    return np.ascontiguousarray(m.reshape(-1))


def build_oracle():
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(ORACLE_DIR, "wm_oracle.c")):
        from winnowmap_amd import build
        with build._Lock(ORACLE_SO):          # (pytest-xdist workers: one make at a time)
            subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)


class EZ(C.Structure):
    _fields_ = [(n, C.c_int32) for n in EZ_FIELDS + ("n_cigar",)]


class M128(C.Structure):
    _fields_ = [("x", C.c_uint64), ("y", C.c_uint64)]


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        build_oracle()
        L = C.CDLL(ORACLE_SO)
        L.wmo_hash64.restype = C.c_uint64
        L.wmo_hash64.argtypes = [C.c_uint64, C.c_uint64]
        L.wmo_fmix64.restype = C.c_uint64
        L.wmo_fmix64.argtypes = [C.c_uint64]
        L.wmo_order.restype = C.c_double
        L.wmo_order.argtypes = [C.c_uint64, C.c_int]
        L.wmo_bloom_new.restype = C.c_void_p
        L.wmo_bloom_new.argtypes = [C.c_uint64]
        L.wmo_bloom_free.argtypes = [C.c_void_p]
        L.wmo_bloom_hash.restype = C.c_uint32
        L.wmo_bloom_hash.argtypes = [C.c_uint64, C.c_uint32]
        L.wmo_bloom_insert.argtypes = [C.c_void_p, C.c_uint64]
        L.wmo_bloom_contains.restype = C.c_int
        L.wmo_bloom_contains.argtypes = [C.c_void_p, C.c_uint64]
        L.wmo_encode_kmer.restype = C.c_uint64
        L.wmo_encode_kmer.argtypes = [C.c_char_p, C.c_int]
        L.wmo_sketch.restype = C.c_int64
        L.wmo_sketch.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p, u64p, u64p, C.c_int64]
        L.wmo_sketch_hpc.restype = C.c_int64
        L.wmo_sketch_hpc.argtypes = L.wmo_sketch.argtypes
        L.wmo_radix_sort_128x.argtypes = [C.c_void_p, C.c_void_p]
        L.wmo_radix_sort_64.argtypes = [C.c_void_p, C.c_void_p]
        L.wmo_chain_dp.restype = C.c_int64
        L.wmo_chain_dp.argtypes = [C.c_int] * 8 + [C.c_float, C.c_int64, C.c_void_p, C.POINTER(C.c_int), u64p, C.c_void_p]
        L.wmo_ksw_extd2.argtypes = [C.c_int, u8p, C.c_int, u8p, C.c_int, i8p] + [C.c_int] * 8 + [C.POINTER(EZ), u32p, C.c_void_p]
        L.wmo_ksw_ll_i16.restype = C.c_int
        L.wmo_ksw_ll_i16.argtypes = [C.c_int, u8p, C.c_int, u8p, C.c_int, i8p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _oracle = L
    return _oracle


class BloomStruct(C.Structure):
    _fields_ = [("table_bits", C.c_uint64), ("salt", C.c_uint32 * 2), ("bits", C.POINTER(C.c_uint8)), ("n_inserted", C.c_uint64)]


def o_bloom(kmers):
    """Build the oracle bloom filter from an iterable of canonical 2-bit k-mers. Returns an opaque handle."""
    L = oracle()
    kmers = list(kmers)
    f = L.wmo_bloom_new(len(kmers))
    for km in kmers:
        L.wmo_bloom_insert(f, int(km))
    return f


def o_bloom_view(f):
    s = C.cast(f, C.POINTER(BloomStruct)).contents
    bits = np.ctypeslib.as_array(s.bits, shape=(s.table_bits // 8,)).copy()
    return int(s.table_bits), (int(s.salt[0]), int(s.salt[1])), bits


def o_sketch(seq, w, k, rid=0, bloom=None, hpc=False):
    L = oracle()
    if isinstance(seq, str):
        seq = seq.encode()
    cap = len(seq) + 8
    ox = np.zeros(cap, np.uint64)
    oy = np.zeros(cap, np.uint64)
    n = (L.wmo_sketch_hpc if hpc else L.wmo_sketch)(seq, len(seq), w, k, rid, bloom, ox, oy, cap)
    return ox[:n].copy(), oy[:n].copy()


def o_radix_sort_128x(x, y):
    L = oracle()
    a = np.empty(len(x), dtype=[("x", np.uint64), ("y", np.uint64)])
    a["x"], a["y"] = x, y
    p = a.ctypes.data
    L.wmo_radix_sort_128x(p, p + 16 * len(a))
    return a["x"].copy(), a["y"].copy()


def o_chain_dp(ax, ay, max_dist_x=5000, min_dist_x=1000, max_dist_y=5000, bw=500, max_skip=25, max_iter=5000,
               min_cnt=3, min_sc=40, gap_scale=1.0, is_cdna=0):
    L = oracle()
    L.wmo_chain_set_cdna(int(is_cdna))
    n = len(ax)
    a = np.empty(n, dtype=[("x", np.uint64), ("y", np.uint64)])
    a["x"], a["y"] = ax, ay
    b = np.empty(max(n, 1), dtype=[("x", np.uint64), ("y", np.uint64)])
    u = np.zeros(max(n, 1), np.uint64)
    n_u = C.c_int(0)
    n_v = L.wmo_chain_dp(max_dist_x, min_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, gap_scale,
                         n, a.ctypes.data, C.byref(n_u), u, b.ctypes.data)
    return u[:n_u.value].copy(), b["x"][:n_v].copy(), b["y"][:n_v].copy()


def o_ksw_extd2(query, target, mat=None, q=4, e=2, q2=24, e2=1, w=751, zdrop=400, end_bonus=-1, flag=0, stats=False):
    L = oracle()
    mat = simple_mat() if mat is None else mat
    query = np.ascontiguousarray(query, np.uint8)
    target = np.ascontiguousarray(target, np.uint8)
    ez = EZ()
    cig = np.zeros(len(query) + len(target) + 4, np.uint32)
    st = (C.c_int * 2)()
    L.wmo_ksw_extd2(len(query), query, len(target), target, 5, mat, q, e, q2, e2, w, zdrop, end_bonus, flag,
                    C.byref(ez), cig, C.cast(st, C.c_void_p) if stats else None)
    d = {n: getattr(ez, n) for n in EZ_FIELDS}
    d["cigar"] = cig[:ez.n_cigar].copy()
    if stats:
        d["stats"] = (st[0], st[1])
    return d


def o_ksw_exts2(query, target, mat=None, q=2, e=1, q2=32, noncan=9, zdrop=200, junc_bonus=9, flag=0, junc=None):
    """oracle restatement of ksw_exts2_sse (splice preset: a=1 b=2 q=2 e=1 q2=32 noncan=9, src/options.c:117-127)"""
    L = oracle()
    L.wmo_ksw_exts2.argtypes = [C.c_int, u8p, C.c_int, u8p, C.c_int, i8p] + [C.c_int] * 7 + [C.c_void_p, C.c_void_p, u32p]
    mat = simple_mat(1, 2, 1) if mat is None else mat
    query = np.ascontiguousarray(query, np.uint8)
    target = np.ascontiguousarray(target, np.uint8)
    ez = EZ()
    cig = np.zeros(len(query) + len(target) + 4, np.uint32)
    jn = None if junc is None else np.ascontiguousarray(junc, np.uint8)
    L.wmo_ksw_exts2(len(query), query, len(target), target, 5, mat, q, e, q2, noncan, zdrop, junc_bonus, flag,
                    None if jn is None else jn.ctypes.data, C.cast(C.byref(ez), C.c_void_p), cig)
    d = {n: getattr(ez, n) for n in EZ_FIELDS}
    d["cigar"] = cig[:ez.n_cigar].copy()
    return d


def o_ksw_ll(query, target, mat=None, gapo=4, gape=2):
    L = oracle()
    mat = simple_mat() if mat is None else mat
    qe, te = C.c_int(), C.c_int()
    s = L.wmo_ksw_ll_i16(len(query), np.ascontiguousarray(query, np.uint8), len(target), np.ascontiguousarray(target, np.uint8),
                         5, mat, gapo, gape, C.byref(qe), C.byref(te))
    return s, qe.value, te.value


# ------------------------------------------------------------------------------------------------
# the real reference (oracle/_ref), optional
# ------------------------------------------------------------------------------------------------
_ref = None


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        L.refshim_idx_build.restype = C.c_void_p
        L.refshim_idx_build.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.refshim_idx_destroy.argtypes = [C.c_void_p]
        L.refshim_idx_nseq.argtypes = [C.c_void_p]
        L.refshim_idx_seqlen.argtypes = [C.c_void_p, C.c_int]
        L.refshim_idx_getseq.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, u8p]
        L.refshim_idx_get.argtypes = [C.c_void_p, C.c_uint64, u64p, C.c_int]
        L.refshim_bloom_contains.argtypes = [C.c_void_p, C.c_uint64]
        L.refshim_bloom_table_bits.restype = C.c_uint64
        L.refshim_bloom_table_bits.argtypes = [C.c_void_p]
        L.refshim_bloom_hash_count.restype = C.c_uint64
        L.refshim_bloom_hash_count.argtypes = [C.c_void_p]
        L.refshim_bloom_table_bytes.restype = C.c_uint64
        L.refshim_bloom_table_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.refshim_sketch.restype = C.c_int64
        L.refshim_sketch.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, u64p, u64p, C.c_int64]
        L.refshim_radix_sort_128x.argtypes = [u64p, u64p, C.c_int64]
        L.refshim_radix_sort_64.argtypes = [u64p, C.c_int64]
        L.refshim_chain_dp.restype = C.c_int64
        L.refshim_chain_dp.argtypes = [C.c_int] * 8 + [C.c_float, C.c_int, C.c_int, C.c_int64, u64p, u64p, C.POINTER(C.c_int), u64p, u64p, u64p]
        L.refshim_ksw_extd2.argtypes = [C.c_int, u8p, C.c_int, u8p, i8p] + [C.c_int] * 8 + [i32p, u32p, C.c_int]
        L.refshim_ksw_extz2.argtypes = [C.c_int, u8p, C.c_int, u8p, i8p] + [C.c_int] * 6 + [i32p, u32p, C.c_int]
        L.refshim_ksw_ll_i16.argtypes = [C.c_int, u8p, C.c_int, u8p, i8p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.refshim_mapopt.restype = C.c_void_p
        L.refshim_mapopt.argtypes = [C.c_char_p, C.c_int64, C.c_void_p]
        L.refshim_preset_k.argtypes = [C.c_char_p]
        L.refshim_preset_w.argtypes = [C.c_char_p]
        L.refshim_map.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, i32p, C.c_int, u32p, C.c_int64, C.POINTER(C.c_int64)]
        _ref = L
    return _ref


def r_sketch(mi, seq, w, k, rid=0, hpc=False):
    L = ref()
    if isinstance(seq, str):
        seq = seq.encode()
    cap = len(seq) + 8
    ox = np.zeros(cap, np.uint64)
    oy = np.zeros(cap, np.uint64)
    n = L.refshim_sketch(mi, seq, len(seq), w, k, rid, 1 if hpc else 0, ox, oy, cap)
    return ox[:n].copy(), oy[:n].copy()


def r_radix_sort_128x(x, y):
    x = np.ascontiguousarray(x, np.uint64).copy()
    y = np.ascontiguousarray(y, np.uint64).copy()
    ref().refshim_radix_sort_128x(x, y, len(x))
    return x, y


def r_chain_dp(ax, ay, max_dist_x=5000, min_dist_x=1000, max_dist_y=5000, bw=500, max_skip=25, max_iter=5000,
               min_cnt=3, min_sc=40, gap_scale=1.0, is_cdna=0):
    n = len(ax)
    ax = np.ascontiguousarray(ax, np.uint64)
    ay = np.ascontiguousarray(ay, np.uint64)
    u = np.zeros(max(n, 1), np.uint64)
    bx = np.zeros(max(n, 1), np.uint64)
    by = np.zeros(max(n, 1), np.uint64)
    n_u = C.c_int(0)
    n_v = ref().refshim_chain_dp(max_dist_x, min_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, gap_scale,
                                 int(is_cdna), 1, n, ax, ay, C.byref(n_u), u, bx, by)
    return u[:n_u.value].copy(), bx[:n_v].copy(), by[:n_v].copy()


def r_ksw_extd2(query, target, mat=None, q=4, e=2, q2=24, e2=1, w=751, zdrop=400, end_bonus=-1, flag=0):
    mat = simple_mat() if mat is None else mat
    query = np.ascontiguousarray(query, np.uint8)
    target = np.ascontiguousarray(target, np.uint8)
    ez = np.zeros(10, np.int32)
    cig = np.zeros(len(query) + len(target) + 4, np.uint32)
    n = ref().refshim_ksw_extd2(len(query), query, len(target), target, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez, cig, len(cig))
    d = {name: int(ez[i]) for i, name in enumerate(EZ_FIELDS)}
    d["cigar"] = cig[:n].copy()
    return d


def r_ksw_exts2(query, target, mat=None, q=2, e=1, q2=32, noncan=9, zdrop=200, junc_bonus=9, flag=0, junc=None):
    R = ref()
    R.refshim_ksw_exts2.argtypes = [C.c_int, u8p, C.c_int, u8p, i8p] + [C.c_int] * 7 + [C.c_void_p, i32p, u32p, C.c_int]
    mat = simple_mat(1, 2, 1) if mat is None else mat
    query = np.ascontiguousarray(query, np.uint8)
    target = np.ascontiguousarray(target, np.uint8)
    ez = np.zeros(10, np.int32)
    cig = np.zeros(len(query) + len(target) + 4, np.uint32)
    jn = None if junc is None else np.ascontiguousarray(junc, np.uint8)
    n = R.refshim_ksw_exts2(len(query), query, len(target), target, mat, q, e, q2, noncan, zdrop, junc_bonus, flag,
                            None if jn is None else jn.ctypes.data, ez, cig, len(cig))
    d = {name: int(ez[i]) for i, name in enumerate(EZ_FIELDS)}
    d["cigar"] = cig[:n].copy()
    return d


def r_ksw_ll(query, target, mat=None, gapo=4, gape=2):
    mat = simple_mat() if mat is None else mat
    qe, te = C.c_int(), C.c_int()
    s = ref().refshim_ksw_ll_i16(len(query), np.ascontiguousarray(query, np.uint8), len(target), np.ascontiguousarray(target, np.uint8),
                                 mat, gapo, gape, C.byref(qe), C.byref(te))
    return s, qe.value, te.value


def cigar_str(cig):
    return "".join("%d%s" % (int(c) >> 4, "MIDN"[int(c) & 0xf]) for c in cig)


def enc(s):
    """ASCII → 0..4 codes (src/sketch.c:19)."""
    t = np.full(256, 4, np.uint8)
    for i, ch in enumerate("ACGT"):
        t[ord(ch)] = i
        t[ord(ch.lower())] = i
    t[ord("U")] = t[ord("u")] = 3
    return t[np.frombuffer(s.encode() if isinstance(s, str) else s, np.uint8)]
