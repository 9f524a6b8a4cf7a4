"""ctypes binding of libwmgpu.so (include/wm_gpu.h). No fallback: if the HIP library is missing or no
device is usable, every call raises."""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WM_LIBWMGPU") or os.path.join(HERE, "libwmgpu.so")      # (WM_LIBWMGPU: a variant build for A/B runs, winnowmap_amd/build.py)


class WmError(RuntimeError):
    pass


class KswScore(C.Structure):
    _fields_ = [("match", C.c_int8), ("mismatch", C.c_int8), ("sc_ambi", C.c_int8),
                ("q", C.c_int8), ("e", C.c_int8), ("q2", C.c_int8), ("e2", C.c_int8)]


KSW_JOB_DTYPE = np.dtype([("q_off", np.uint32), ("t_off", np.uint32), ("qlen", np.int32), ("tlen", np.int32),
                          ("w", np.int32), ("zdrop", np.int32), ("end_bonus", np.int32), ("flag", np.int32)])
KSW_RES_DTYPE = np.dtype([("max", np.int32), ("zdropped", np.int32), ("max_q", np.int32), ("max_t", np.int32),
                          ("mqe", np.int32), ("mqe_t", np.int32), ("mte", np.int32), ("mte_q", np.int32),
                          ("score", np.int32), ("reach_end", np.int32), ("n_cigar", np.int32), ("cig_off", np.uint32)])

KSW_F_ZDWALK = 0x10000        # wm_ksw_batch_pos_zd: run the z-drop scan on this job's alignment
KSW_POS_DTYPE = np.dtype([("qwin_off", np.int64), ("qwin_len", np.int32), ("q_pos", np.int32), ("rid", np.int32), ("t_pos", np.int32),
                          ("qlen", np.int32), ("tlen", np.int32), ("w", np.int32), ("zdrop", np.int32), ("end_bonus", np.int32), ("flag", np.int32),
                          ("step", np.int8), ("has_n", np.int8), ("pad", np.int8, (6,))])      # wm_ksw_pos_t

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WmError("libwmgpu.so is not built (run `python -m winnowmap_amd.build`); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.wm_last_error.restype = C.c_char_p
        L.wm_ctx_create.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
        L.wm_ctx_destroy.argtypes = [C.c_void_p]
        L.wm_last_kernel_ms.restype = C.c_float
        L.wm_last_kernel_ms.argtypes = [C.c_void_p]
        L.wm_ksw_batch.argtypes = [C.c_void_p, C.POINTER(KswScore), C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                   C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.wm_ksw_dev_prepare.argtypes = [C.c_void_p, C.POINTER(KswScore), C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.wm_ksw_dev_run.argtypes = [C.c_void_p, C.c_void_p]
        L.wm_ksw_dev_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.wm_ksw_dev_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.wm_ksw_dev_free.argtypes = [C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise WmError("libwmgpu error %d: %s" % (rc, lib().wm_last_error().decode()))


class Context:
    """One GPU context (HIP device + stream + HBM arena)."""

    def __init__(self, device=0, arena_bytes=0):
        self._h = C.c_void_p()
        _chk(lib().wm_ctx_create(device, arena_bytes, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().wm_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- ksw -------------------------------------------------------------------------------------
    def ksw_batch(self, score, jobs, seqs):
        """jobs: structured array KSW_JOB_DTYPE; seqs: uint8 codes. Returns (results KSW_RES_DTYPE, cigar pool)."""
        jobs = np.ascontiguousarray(jobs, KSW_JOB_DTYPE)
        seqs = np.ascontiguousarray(seqs, np.uint8)
        res = np.zeros(len(jobs), KSW_RES_DTYPE)
        cap = int((jobs["qlen"].astype(np.int64) + jobs["tlen"] + 2).sum()) + 16
        pool = np.zeros(cap, np.uint32)
        used = C.c_size_t(0)
        _chk(lib().wm_ksw_batch(self._h, C.byref(score), len(jobs), jobs.ctypes.data, seqs.ctypes.data, seqs.nbytes,
                                res.ctypes.data, pool.ctypes.data, cap, C.byref(used)))
        return res, pool[:used.value]

    def ksw_exts2_batch(self, score, noncan, junc_bonus, jobs, seqs, junc=None):
        """wm_ksw_exts2_batch: the splice-aware extension (ksw_exts2_sse); junc = None or uint8 array parallel to seqs"""
        jobs = np.ascontiguousarray(jobs, KSW_JOB_DTYPE)
        seqs = np.ascontiguousarray(seqs, np.uint8)
        jn = None if junc is None else np.ascontiguousarray(junc, np.uint8)
        assert jn is None or jn.nbytes == seqs.nbytes
        res = np.zeros(len(jobs), KSW_RES_DTYPE)
        cap = int((jobs["qlen"].astype(np.int64) + jobs["tlen"] + 2).sum()) + 16
        pool = np.zeros(cap, np.uint32)
        used = C.c_size_t(0)
        lib().wm_ksw_exts2_batch.argtypes = [C.c_void_p, C.POINTER(KswScore), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        _chk(lib().wm_ksw_exts2_batch(self._h, C.byref(score), noncan, junc_bonus, len(jobs), jobs.ctypes.data, seqs.ctypes.data, seqs.nbytes,
                                      None if jn is None else jn.ctypes.data, res.ctypes.data, pool.ctypes.data, cap, C.byref(used)))
        return res, pool[:used.value]

    def reads_upload(self, codes):
        """0..4 codes of the current mini-batch, kept resident for position jobs (wm_reads_upload)"""
        codes = np.ascontiguousarray(codes, np.uint8)
        lib().wm_reads_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        _chk(lib().wm_reads_upload(self._h, codes.ctypes.data, codes.nbytes))

    def ksw_batch_pos(self, score, jobs):
        """jobs: structured array KSW_POS_DTYPE (operands as positions in the resident reads / packed reference, wm_ksw_batch_pos)"""
        jobs = np.ascontiguousarray(jobs, KSW_POS_DTYPE)
        assert KSW_POS_DTYPE.itemsize == 56
        res = np.zeros(len(jobs), KSW_RES_DTYPE)
        cap = int((np.maximum(jobs["qlen"], 0).astype(np.int64) + np.maximum(jobs["tlen"], 0) + 2).sum()) + 16
        pool = np.zeros(cap, np.uint32)
        used = C.c_size_t(0)
        lib().wm_ksw_batch_pos.argtypes = [C.c_void_p, C.POINTER(KswScore), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        _chk(lib().wm_ksw_batch_pos(self._h, C.byref(score), len(jobs), jobs.ctypes.data, res.ctypes.data, pool.ctypes.data, cap, C.byref(used)))
        return res, pool[:used.value]

    def ksw_batch_pos_zd(self, score, jobs):
        """wm_ksw_batch_pos_zd: as ksw_batch_pos, plus the z-drop scan (mm_test_zdrop's walk, src/align.c:32-66) of every job whose flag carries
        KSW_F_ZDWALK, run on the device over the finished CIGARs -> (results, pool, zd[n, 5] = max_zdrop, t0, t1, q0, q1)"""
        jobs = np.ascontiguousarray(jobs, KSW_POS_DTYPE)
        res = np.zeros(len(jobs), KSW_RES_DTYPE)
        cap = int((np.maximum(jobs["qlen"], 0).astype(np.int64) + np.maximum(jobs["tlen"], 0) + 2).sum()) + 16
        pool = np.zeros(cap, np.uint32)
        zd = np.zeros((len(jobs), 5), np.int32)
        used = C.c_size_t(0)
        lib().wm_ksw_batch_pos_zd.argtypes = [C.c_void_p, C.POINTER(KswScore), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]
        _chk(lib().wm_ksw_batch_pos_zd(self._h, C.byref(score), len(jobs), jobs.ctypes.data, res.ctypes.data, pool.ctypes.data, cap, C.byref(used), zd.ctypes.data))
        return res, pool[:used.value], zd

    def ksw_prepare(self, score, jobs, seqs):
        jobs = np.ascontiguousarray(jobs, KSW_JOB_DTYPE)
        seqs = np.ascontiguousarray(seqs, np.uint8)
        h = C.c_void_p()
        _chk(lib().wm_ksw_dev_prepare(self._h, C.byref(score), len(jobs), jobs.ctypes.data, seqs.ctypes.data, seqs.nbytes, C.byref(h)))
        return KswDevBatch(self, h, jobs)


class KswDevBatch:
    def __init__(self, ctx, h, jobs):
        self.ctx, self._h, self.jobs = ctx, h, jobs

    def run(self):
        _chk(lib().wm_ksw_dev_run(self.ctx._h, self._h))

    def stats(self):
        cells, tbb, dp, bt = C.c_uint64(), C.c_uint64(), C.c_float(), C.c_float()
        lib().wm_ksw_dev_stats(self._h, C.byref(cells), C.byref(tbb), C.byref(dp), C.byref(bt))
        return dict(cells=cells.value, tb_bytes=tbb.value, dp_ms=dp.value, bt_ms=bt.value)

    def fetch(self):
        res = np.zeros(len(self.jobs), KSW_RES_DTYPE)
        cap = int((self.jobs["qlen"].astype(np.int64) + self.jobs["tlen"] + 2).sum()) + 16
        pool = np.zeros(cap, np.uint32)
        used = C.c_size_t(0)
        _chk(lib().wm_ksw_dev_fetch(self.ctx._h, self._h, res.ctypes.data, pool.ctypes.data, cap, C.byref(used)))
        return res, pool[:used.value]

    def free(self):
        if self._h:
            lib().wm_ksw_dev_free(self.ctx._h, self._h)
            self._h = None


def set_ksw_routing(on=-1, rows4=-1, rows8=-1):
    """wm_ksw_set_routing: which alignments run on the stripe-pipelined multi-wave kernels (results never depend on it)"""
    lib().wm_ksw_set_routing.argtypes = [C.c_int, C.c_int, C.c_int]
    lib().wm_ksw_set_routing.restype = None
    lib().wm_ksw_set_routing(on, rows4, rows8)


def set_ksw_chain_routing(mode=-1, min_rows_exact=-1, bp=-1):
    """wm_ksw_set_chain_routing: which alignments run on the chained-workgroup kernels (results never depend on it)"""
    lib().wm_ksw_set_chain_routing.argtypes = [C.c_int, C.c_int, C.c_int]
    lib().wm_ksw_set_chain_routing.restype = None
    lib().wm_ksw_set_chain_routing(mode, min_rows_exact, bp)


def set_ksw_dual(on):
    """wm_ksw_set_dual: two alignments per wavefront for the gap-fill classes (results never depend on it)"""
    lib().wm_ksw_set_dual.argtypes = [C.c_int]
    lib().wm_ksw_set_dual.restype = None
    lib().wm_ksw_set_dual(int(on))


def ksw_dual_enabled():
    lib().wm_ksw_dual_enabled.restype = C.c_int
    return bool(lib().wm_ksw_dual_enabled())


def build_defines():
    """the kernel-variant defines the loaded library was compiled with (wm_build_defines)"""
    lib().wm_build_defines.restype = C.c_char_p
    return lib().wm_build_defines().decode()


STRIPE_TIMING_FIELDS = ("scan", "epoch", "cells", "wait_left", "book", "wait_right", "publish", "rows", "epochs", "total", "waves")


def stripe_timing(reset=True):
    """per-phase shader-clock cycles of the stripe-pipelined ksw kernel, summed over wavefronts since the last reset (wm_debug_stripe_timing; only in a
    library built with WM_KERNEL_DEFINES="WM_STRIPE_TIMING=1", WmError otherwise)"""
    out = np.zeros(16, np.uint64)
    lib().wm_debug_stripe_timing.argtypes = [C.c_void_p, C.c_int]
    _chk(lib().wm_debug_stripe_timing(out.ctypes.data, 1 if reset else 0))
    return {k: int(out[i]) for i, k in enumerate(STRIPE_TIMING_FIELDS)}


def pack_jobs(pairs, w=751, zdrop=400, end_bonus=-1, flag=0):
    """pairs: list of (query codes, target codes[, dict overrides]). Returns (jobs, seqs)."""
    jobs = np.zeros(len(pairs), KSW_JOB_DTYPE)
    chunks, off = [], 0
    for i, p in enumerate(pairs):
        q, t = np.asarray(p[0], np.uint8), np.asarray(p[1], np.uint8)
        o = dict(w=w, zdrop=zdrop, end_bonus=end_bonus, flag=flag)
        if len(p) > 2:
            o.update(p[2])
        jobs[i] = (off, off + len(q), len(q), len(t), o["w"], o["zdrop"], o["end_bonus"], o["flag"])
        chunks += [q, t]
        off += len(q) + len(t)
    seqs = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
    return jobs, seqs


# ---- index + mapper ------------------------------------------------------------------------------------
def _bind_map(L):
    if getattr(L, "_wm_map_bound", False):
        return
    L.wm_index_build.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.wm_index_destroy.argtypes = [C.c_void_p]
    L.wm_index_upload.argtypes = [C.c_void_p, C.c_void_p]
    L.wm_index_n_seq.argtypes = [C.c_void_p]
    L.wm_index_seq_name.restype = C.c_char_p
    L.wm_index_seq_name.argtypes = [C.c_void_p, C.c_int]
    L.wm_index_seq_len.argtypes = [C.c_void_p, C.c_int]
    L.wm_index_n_minimizers.restype = C.c_uint64
    L.wm_index_n_minimizers.argtypes = [C.c_void_p]
    L.wm_mapper_create.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(C.c_void_p)]
    L.wm_mapper_destroy.argtypes = [C.c_void_p]
    L.wm_mapper_set_threads.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    L.wm_map_reads.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_void_p,
                               C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.wm_map_reads_slot.argtypes = [C.c_void_p, C.c_int] + list(L.wm_map_reads.argtypes[1:])
    L.wm_mapper_stats.argtypes = [C.c_void_p, C.c_void_p]
    L.wm_sam_header.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_size_t)]
    L.wm_last_aux_ms.restype = C.c_float
    L.wm_last_aux_ms.argtypes = [C.c_void_p]
    L.wm_write_repetitive_kmers.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_char_p, C.POINTER(C.c_uint64)]
    L.wm_index_export.argtypes = [C.c_void_p] + [C.c_void_p] * 8
    L.wm_index_import.argtypes = [C.c_void_p] * 8 + [C.POINTER(C.c_void_p)]
    L._wm_map_bound = True


def write_repetitive_kmers(fasta, k, out_path, distinct=0.9998):
    """Write the -W list of `fasta` (see include/wm_gpu.h). Returns the number of k-mers written."""
    L = lib()
    _bind_map(L)
    n = C.c_uint64()
    _chk(L.wm_write_repetitive_kmers(os.fsencode(fasta), k, distinct, os.fsencode(out_path), C.byref(n)))
    return int(n.value)


def write_repetitive_kmers_gpu(ctx, fasta, k, out_path, distinct=0.9998):
    """The -W list counted on the device (wm_write_repetitive_kmers_gpu). Returns (number of k-mers written, stats dict)."""
    L = lib()
    L.wm_write_repetitive_kmers_gpu.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_double, C.c_char_p, C.POINTER(C.c_uint64), C.c_void_p]
    n = C.c_uint64()
    st = np.zeros(4, np.float64)
    _chk(L.wm_write_repetitive_kmers_gpu(ctx._h, os.fsencode(fasta), k, distinct, os.fsencode(out_path), C.byref(n), st.ctypes.data))
    return int(n.value), {"read_encode_s": float(st[0]), "device_s": float(st[1]), "write_s": float(st[2]), "distinct_kmers": int(st[3])}


_EXPORT_DTYPES = (np.uint32, np.uint64, np.uint64, np.uint64, np.uint8, np.uint64, np.uint8)


class MapOpt(C.Structure):
    """wm_mapopt_t (include/wm_gpu.h) = the mirrored fields of mm_mapopt_t, in that order"""
    _fields_ = [("flag", C.c_int64)] + [(n, C.c_int32) for n in ("seed", "sdust_thres", "max_qlen", "bw", "max_gap", "max_gap_ref", "min_gap_ref", "max_frag_len",
                                                                  "max_chain_skip", "max_chain_iter", "min_cnt", "min_chain_score")] + \
               [("chain_gap_scale", C.c_float)] + [(n, C.c_int32) for n in ("SVaware", "SVawareMinReadLength", "suffixSampleOffset", "min_mapq")] + [("min_qcov", C.c_float)] + \
               [(n, C.c_int32) for n in ("minPrefixLength", "maxPrefixLength")] + [("prefixIncrementFactor", C.c_float)] + \
               [(n, C.c_int32) for n in ("stage2_bw", "stage2_zdrop_inv", "stage2_max_gap")] + [("mask_level", C.c_float), ("mask_len", C.c_int32), ("pri_ratio", C.c_float), ("best_n", C.c_int32)] + \
               [(n, C.c_int32) for n in ("max_join_long", "max_join_short", "min_join_flank_sc")] + [("min_join_flank_ratio", C.c_float), ("alt_drop", C.c_float)] + \
               [(n, C.c_int32) for n in ("a", "b", "q", "e", "q2", "e2", "sc_ambi", "zdrop", "zdrop_inv", "end_bonus", "min_dp_max", "min_ksw_len")] + \
               [("max_clip_ratio", C.c_float), ("mid_occ_frac", C.c_float)] + [(n, C.c_int32) for n in ("min_mid_occ", "mid_occ", "max_occ")] + [("mini_batch_size", C.c_int64), ("max_sw_mat", C.c_int64)] + \
               [(n, C.c_int32) for n in ("noncan", "junc_bonus", "anchor_ext_len", "anchor_ext_shift")]


def mapopt_preset(preset):
    """wm_mapopt_preset = mm_set_opt(0) + mm_set_opt(preset): -> (MapOpt, k, w)"""
    L = lib()
    L.wm_mapopt_preset.argtypes = [C.c_char_p, C.POINTER(MapOpt), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    o = MapOpt(); k = C.c_int(); w = C.c_int()
    _chk(L.wm_mapopt_preset((preset or "").encode(), C.byref(o), C.byref(k), C.byref(w)))
    return o, k.value, w.value


def build_index_parts(fasta, kmer_file, k, w, batch_bases, n_threads=8):
    """wm_index_build_parts: the reference indexed in parts of `batch_bases` (the reference's -I) -> list of Index"""
    L = lib()
    _bind_map(L)
    L.wm_index_build_parts.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
    cap = 4096
    arr = (C.c_void_p * cap)()
    n = C.c_int()
    _chk(L.wm_index_build_parts(os.fsencode(fasta), os.fsencode(kmer_file) if kmer_file else None, k, w, n_threads, batch_bases, arr, cap, C.byref(n)))
    return [Index(_handle=C.c_void_p(arr[i])) for i in range(n.value)]


def map_file_multi(mappers, reads_path, out_path, mini_batch_bases=0):
    """wm_map_file_multi: the file loop over several mappers (one per GPU), two lanes each, records in input order"""
    L = lib()
    L.wm_map_file_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_char_p, C.c_int64, C.c_void_p]
    arr = (C.c_void_p * len(mappers))(*[m._h for m in mappers])
    st = np.zeros(6, np.float64)
    _chk(L.wm_map_file_multi(arr, len(mappers), os.fsencode(reads_path), os.fsencode(out_path), mini_batch_bases, st.ctypes.data))
    return dict(zip(("reads", "bases", "batches", "t_read", "t_map", "t_write"), (float(x) for x in st)))


def map_file_split(ctx, parts, opt, n_threads, reads_path, out_path, mini_batch_bases=0):
    """wm_map_file_split: the reads against every index part in turn, hits merged like `--split-prefix` (mm_split_merge)"""
    L = lib()
    L.wm_map_file_split.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(MapOpt), C.c_int, C.c_char_p, C.c_char_p, C.c_int64, C.c_void_p]
    arr = (C.c_void_p * len(parts))(*[p._h for p in parts])
    st = np.zeros(6, np.float64)
    _chk(L.wm_map_file_split(ctx._h, len(parts), arr, C.byref(opt), n_threads, os.fsencode(reads_path), os.fsencode(out_path), mini_batch_bases, st.ctypes.data))
    return dict(zip(("reads", "bases", "batches", "t_read", "t_map", "t_write"), (float(x) for x in st)))


def map_file_split_fasta(ctx, fasta, kmer_file, k, w, batch_bases, opt, n_threads, reads_path, out_path, on_device=False, build_threads=8, mini_batch_bases=0):
    """wm_map_file_split_fasta: like `winnowmap -I <batch_bases> --split-prefix`, one index part in memory at a time (built when its turn comes)"""
    L = lib()
    L.wm_map_file_split_fasta.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.POINTER(MapOpt), C.c_int,
                                          C.c_char_p, C.c_char_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int)]
    st = np.zeros(6, np.float64)
    n = C.c_int()
    _chk(L.wm_map_file_split_fasta(ctx._h, os.fsencode(fasta), os.fsencode(kmer_file) if kmer_file else None, k, w, build_threads, batch_bases, 1 if on_device else 0,
                                   C.byref(opt), n_threads, os.fsencode(reads_path), os.fsencode(out_path), mini_batch_bases, st.ctypes.data, C.byref(n)))
    d = dict(zip(("reads", "bases", "batches", "t_read", "t_map", "t_write"), (float(x) for x in st)))
    d["parts"] = n.value
    return d


class SplitRun:
    """wm_split_begin / wm_split_add_part / wm_split_finish: a reference indexed in parts, one part at a time (src/main.c:398-429)"""

    def __init__(self, ctx, opt, k, w, n_threads, reads_path, mini_batch_bases=0):
        L = lib()
        L.wm_split_begin.argtypes = [C.c_void_p, C.POINTER(MapOpt), C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int64, C.POINTER(C.c_void_p)]
        L.wm_split_add_part.argtypes = [C.c_void_p, C.c_void_p]
        L.wm_split_finish.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        L.wm_split_abort.argtypes = [C.c_void_p]
        L.wm_split_abort.restype = None
        self._h = C.c_void_p()
        _chk(L.wm_split_begin(ctx._h, C.byref(opt), k, w, n_threads, os.fsencode(reads_path), mini_batch_bases, C.byref(self._h)))

    def add_part(self, idx):
        _chk(lib().wm_split_add_part(self._h, idx._h))

    def finish(self, out_path):
        st = np.zeros(6, np.float64)
        h, self._h = self._h, None
        _chk(lib().wm_split_finish(h, os.fsencode(out_path), st.ctypes.data))
        return dict(zip(("reads", "bases", "batches", "t_read", "t_map", "t_write"), (float(x) for x in st)))

    def abort(self):
        if self._h:
            lib().wm_split_abort(self._h)
            self._h = None


class Index:
    """Reference index (host build, mm_idx_gen semantics); `upload(ctx)` copies the flat arrays to HBM."""

    def __init__(self, fasta=None, kmer_file=None, k=15, w=50, n_threads=8, _handle=None, hpc=False):
        L = lib()
        _bind_map(L)
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
            return
        L.wm_index_build_flag.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        _chk(L.wm_index_build_flag(os.fsencode(fasta), os.fsencode(kmer_file) if kmer_file else None, k, w, 1 if hpc else 0, n_threads, C.byref(self._h)))      # hpc: MM_I_HPC (-H)

    @staticmethod
    def build_on_device(ctx, fasta, kmer_file=None, k=15, w=50, n_threads=8, hpc=False):
        """wm_index_build_gpu[_flag]: the reference is sketched on the device (one wavefront per contig). Returns (Index, stats dict)."""
        L = lib()
        _bind_map(L)
        L.wm_index_build_gpu_flag.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]
        h = C.c_void_p()
        st = np.zeros(4, np.float64)
        _chk(L.wm_index_build_gpu_flag(ctx._h, os.fsencode(fasta), os.fsencode(kmer_file) if kmer_file else None, k, w, 1 if hpc else 0, n_threads, C.byref(h), st.ctypes.data))
        L.wm_last_aux_ms.restype = C.c_float
        L.wm_last_aux_ms.argtypes = [C.c_void_p]
        return Index(_handle=h), {"read_pack_s": float(st[0]), "device_sketch_s": float(st[1]), "table_s": float(st[2]), "minimizers": int(st[3]),
                                  "table_on_device_s": float(L.wm_last_aux_ms(ctx._h)) * 1e-3}       # (< 0: the table was built by the host)

    def read_junc_bed(self, path):
        """--junc-bed: annotated introns for splice mode's junction bonus (wm_index_read_junc_bed = mm_idx_bed_read, src/index.c:756)"""
        L = lib()
        L.wm_index_read_junc_bed.argtypes = [C.c_void_p, C.c_char_p]
        _chk(L.wm_index_read_junc_bed(self._h, os.fsencode(path)))

    def export_arrays(self):
        """(sizes9, [S, hkey, hval, P, bloom, seq_meta, names]) as numpy arrays — the payload of the RCCL broadcast."""
        L = lib()
        sizes = np.zeros(9, np.uint64)
        _chk(L.wm_index_export(self._h, sizes.ctypes.data, None, None, None, None, None, None, None))
        n = [int(sizes[0]), int(sizes[1]), int(sizes[1]), int(sizes[2]), int(sizes[3]), 2 * int(sizes[4]), int(sizes[5])]
        arrs = [np.zeros(max(m, 1), dt) for m, dt in zip(n, _EXPORT_DTYPES)]
        _chk(L.wm_index_export(self._h, sizes.ctypes.data, *[a.ctypes.data for a in arrs]))
        return sizes, arrs

    @staticmethod
    def from_arrays(sizes, arrs):
        L = lib()
        _bind_map(L)
        h = C.c_void_p()
        arrs = [np.ascontiguousarray(a, dt) for a, dt in zip(arrs, _EXPORT_DTYPES)]
        sizes = np.ascontiguousarray(sizes, np.uint64)
        _chk(L.wm_index_import(sizes.ctypes.data, *[a.ctypes.data for a in arrs], C.byref(h)))
        return Index(_handle=h)

    def save(self, path):
        """write the reference's MMI index file format"""
        L = lib()
        L.wm_index_save.argtypes = [C.c_void_p, C.c_char_p]
        _chk(L.wm_index_save(self._h, path.encode()))

    @staticmethod
    def load(path, kmer_file=None):
        L = lib()
        _bind_map(L)
        L.wm_index_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        h = C.c_void_p()
        _chk(L.wm_index_load(path.encode(), kmer_file.encode() if kmer_file else None, C.byref(h)))
        return Index(_handle=h)

    def upload(self, ctx):
        _chk(lib().wm_index_upload(ctx._h, self._h))

    def upload_dev(self, ctx, d_ptrs, src_device):
        """wm_index_upload_dev: the flat arrays S, hkey, hval, P, bloom from DEVICE pointers (e.g. the receive buffers of the RCCL broadcast) on src_device"""
        L = lib()
        L.wm_index_upload_dev.argtypes = [C.c_void_p, C.c_void_p] + [C.c_void_p] * 5 + [C.c_int]
        _chk(L.wm_index_upload_dev(ctx._h, self._h, *[C.c_void_p(int(p)) for p in d_ptrs], int(src_device)))

    def upload_peer(self, dst_ctx, src_ctx):
        """wm_index_upload_peer: the copy context `src_ctx` holds goes to `dst_ctx` (any GPU of the node), device to device"""
        L = lib()
        L.wm_index_upload_peer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _chk(L.wm_index_upload_peer(dst_ctx._h, self._h, src_ctx._h))

    @property
    def n_minimizers(self):
        if not self._h:
            raise RuntimeError("the index has been closed")
        return int(lib().wm_index_n_minimizers(self._h))

    def names(self):
        L = lib()
        return [L.wm_index_seq_name(self._h, i).decode() for i in range(L.wm_index_n_seq(self._h))]

    def close(self):
        if self._h:
            lib().wm_index_destroy(self._h)
            self._h = C.c_void_p()


MM_F_CIGAR, MM_F_OUT_SAM, MM_F_OUT_CG = 0x4, 0x8, 0x20
STAT_NAMES = ("super_steps", "ksw_jobs", "chain_jobs", "seed_jobs", "sketch_jobs", "dp_cells", "ksw_kernel_us", "aux_kernel_us", "read_bases")


class Mapper:
    """The batched replacement of kt_for(worker_for) (src/map.c:1164): maps a list of reads in one call."""

    def __init__(self, ctx, index, preset="map-ont", flag=MM_F_CIGAR | MM_F_OUT_CG):
        L = lib()
        _bind_map(L)
        self.ctx, self.index = ctx, index
        self._h = C.c_void_p()
        _chk(L.wm_mapper_create(ctx._h, index._h, preset.encode() if preset else None, flag, C.byref(self._h)))

    def set_threads(self, n_threads, arena_bytes_per_thread=0):
        _chk(lib().wm_mapper_set_threads(self._h, n_threads, arena_bytes_per_thread))

    @staticmethod
    def marshal(names, seqs):
        """the C arrays wm_map_reads takes (name pointers, sequence pointers, lengths): building them from Python lists costs tens of
        milliseconds per 10^5 reads, which a caller timing the mapper does once, outside its clock"""
        n = len(seqs)
        nm = (C.c_char_p * n)(*[x if isinstance(x, bytes) else x.encode() for x in names])
        sq = (C.c_char_p * n)(*seqs)
        lens = np.array([len(s) for s in seqs], np.int32)
        return n, nm, sq, lens, (names, seqs)      # (the lists keep the pointed-to bytes alive)

    def map(self, names, seqs=None, copy_text=True, slot=0):
        """names: list of str/bytes; seqs: list of bytes (ASCII) — or names = the tuple Mapper.marshal() returned. Returns
        (text, hits[n_hits,16], cigars, first[n+1]).
        copy_text=False returns the text LENGTH instead of a Python copy of the records (they stay in the library's buffer).
        slot (0 .. WM_MAX_SLOTS - 1 = 3): calls on different slots may run concurrently from different threads (wm_map_reads_slot; ctypes releases the GIL)."""
        L = lib()
        n, nm, sq, lens, _keep = names if seqs is None else Mapper.marshal(names, seqs)
        text, tlen = C.c_char_p(), C.c_size_t()
        hits, cig, first = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _chk(L.wm_map_reads_slot(self._h, slot, n, nm, sq, lens.ctypes.data, C.byref(text), C.byref(tlen), C.byref(hits), C.byref(cig), C.byref(first)))
        fa = np.ctypeslib.as_array(C.cast(first, C.POINTER(C.c_int64)), shape=(n + 1,)).copy()
        nh = int(fa[n])
        ha = np.ctypeslib.as_array(C.cast(hits, C.POINTER(C.c_int32)), shape=(nh, 16)).copy() if nh else np.zeros((0, 16), np.int32)
        if not copy_text:
            return tlen.value, ha, None, fa
        nc = int(ha[:, 7].sum()) if nh else 0
        ca = np.ctypeslib.as_array(C.cast(cig, C.POINTER(C.c_uint32)), shape=(nc,)).copy() if nc else np.zeros(0, np.uint32)
        return C.string_at(text, tlen.value), ha, ca, fa

    def rep_len_defined(self, slot=0):
        """uint8 per read of the slot's last map() call: 1 where rep_len (hence MAPQ and rl:i) is defined by the reference itself (wm_map_reads_rep_len_defined)"""
        L = lib()
        L.wm_map_reads_rep_len_defined.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        p, n = C.c_void_p(), C.c_size_t()
        _chk(L.wm_map_reads_rep_len_defined(self._h, slot, C.byref(p), C.byref(n)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint8)

    def stats(self):
        a = np.zeros(9, np.uint64)
        lib().wm_mapper_stats(self._h, a.ctypes.data)
        return dict(zip(STAT_NAMES, (int(x) for x in a)))

    def map_file(self, reads_path, out_path, mini_batch_bases=0):
        """FASTA/FASTQ(.gz) -> PAF/SAM file, mini-batch pipeline (wm_map_file). Returns the stats dict."""
        L = lib()
        L.wm_map_file.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int64, C.c_void_p]
        st = np.zeros(6, np.float64)
        _chk(L.wm_map_file(self._h, reads_path.encode(), out_path.encode(), mini_batch_bases, st.ctypes.data))
        return dict(zip(("reads", "bases", "batches", "t_read", "t_map", "t_write"), (float(x) for x in st)))

    def kernel_stats(self):
        """per ksw kernel class: dict class -> (ms, cells, launches)"""
        out = np.zeros(3 * 64, np.float64)
        n = C.c_int()
        lib().wm_mapper_kernel_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        _chk(lib().wm_mapper_kernel_stats(self._h, out.ctypes.data, len(out), C.byref(n)))
        return {k: (float(out[3 * k]), float(out[3 * k + 1]), int(out[3 * k + 2])) for k in range(n.value)}

    def kernel_union(self, since_ms=0.0):
        """(dict class -> ms with at least one launch of the class running since `since_ms`, device clock now in ms)"""
        ncl = int(lib().wm_ksw_n_classes())
        out = np.zeros(max(64, ncl), np.float64)
        now = C.c_double()
        lib().wm_mapper_kernel_union.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        _chk(lib().wm_mapper_kernel_union(self._h, float(since_ms), out.ctypes.data, len(out), C.byref(now)))
        return {k: float(out[k]) for k in range(ncl)}, now.value

    def host_stats(self):
        """host time accounting since the mapper was created (wm_mapper_host_stats), seconds summed over the worker threads"""
        a = np.zeros(24, np.float64)
        lib().wm_mapper_host_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _chk(lib().wm_mapper_host_stats(self._h, a.ctypes.data, len(a)))
        ops = ("window", "seed", "chain", "ksw")     # (the mapper issues fused window calls and ksw calls; the per-stage slots stay 0)
        return {"cpu_glue_s": float(a[0]), "idle_wall_s": float(a[1]), "cpu_batched_s": {o: float(a[2 + i]) for i, o in enumerate(ops)},
                "wall_batched_s": {o: float(a[6 + i]) for i, o in enumerate(ops)}, "batched_calls": {o: int(a[10 + i]) for i, o in enumerate(ops)},
                "map_wall_s": float(a[14]), "format_wall_s": float(a[15]), "threads": int(a[16]), "cpu_help_s": float(a[17]),
                "glue_wall_s": float(a[18]), "lock_wait_wall_s": float(a[19]), "workers_wall_s": float(a[20])}

    def close(self):
        if self._h:
            lib().wm_mapper_destroy(self._h)
            self._h = C.c_void_p()
