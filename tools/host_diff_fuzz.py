"""Differential fuzz of the HOST side of the path (MCAS glue, hit.c / align.c restatement, splice mode) against the reference's library:
the product's host mapper on oracle-backed device operations (tests/host_harness) vs refshim_map (oracle/_ref), hit by hit and CIGAR by CIGAR,
over random references / reads / presets.   python tools/host_diff_fuzz.py [first_seed] [n_seeds]"""
import ctypes as C, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import wmtest as W
from winnowmap_amd import build, synth

# (preset, k, w, read profile, -W list, -H: homopolymer-compressed index, src/sketch.c:152-163 + mm_adjust_minier's HPC branch)
CASES = [("map-ont", 15, 50, "ont", True, 0), ("map-pb", 15, 50, "hifi", True, 0), ("map-ont", 15, 50, "ont", False, 0), ("asm20", 19, 50, "hifi", False, 0),
         ("splice", 15, 25, None, False, 0), ("map-pb-clr", 15, 50, "ont", False, 0), ("map-ont", 15, 50, "ont", True, 1), ("map-pb", 19, 10, "hifi", False, 1)]


def run(s0, ns, verbose=True):
    """-> (reads compared, mismatching reads, reads whose MAPQ was part of the comparison)"""
    H = C.CDLL(build.build_harness())
    H.h_index_build_flag.restype = C.c_void_p
    H.h_index_build_flag.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
    H.h_map.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_char_p, C.c_int, C.c_char_p, W.i32p, C.c_int, W.u32p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]
    R = W.ref()
    R.refshim_idx_build_flag.restype = C.c_void_p
    R.refshim_idx_build_flag.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
    reads_total = bad = with_mapq = 0
    for seed in range(s0, s0 + ns):
        preset, k, w, prof, use_w, hpc = CASES[seed % len(CASES)]
        rng = np.random.default_rng(seed)
        tmp = tempfile.mkdtemp()
        ref = synth.make_reference(int(rng.integers(1, 4)), int(rng.integers(150000, 400000)), seed, repeat_frac=float(rng.choice([0.0, 0.05, 0.15])))
        if preset == "splice":
            reads = synth.make_transcripts(ref, 12, seed + 1)
        else:
            rl = int(rng.choice([1500, 6000, 12000, 25000, 45000]))
            rl = min(rl, min(len(c) for c in ref) - 2000)
            reads, _ = synth.make_reads(ref, 8, rl, seed + 1, profile=prof, sv_frac=float(rng.choice([0.0, 0.3, 0.6])))
            # hard cases: ambiguous bases, a read made of two distant pieces, a very short read, pure noise, a read with a long insertion
            r0 = reads[0].copy(); p0 = int(rng.integers(0, max(1, len(r0) - 40))); r0[p0:p0 + int(rng.integers(1, 30))] = 4
            c = ref[0]
            a0, b0 = int(rng.integers(0, len(c) // 2 - 5000)), int(rng.integers(len(c) // 2, len(c) - 5000))
            chim = np.concatenate([c[a0:a0 + 4000], synth.revcomp_codes(c[b0:b0 + 4000])])
            ins = np.concatenate([c[a0:a0 + 3000], synth.random_codes(int(rng.integers(50, 2500)), rng), c[a0 + 3000:a0 + 7000]])
            reads += [r0, synth.mutate_codes(chim, rng, 0.03, 0.02, 0.02), c[a0:a0 + int(rng.integers(20, 300))].copy(), synth.random_codes(3000, rng),
                      synth.mutate_codes(ins, rng, 0.03, 0.02, 0.02)]
        fa = tmp + "/ref.fa"
        synth.write_fasta(fa, ref)
        kf = b""
        if use_w:
            km, cnt = synth.repetitive_kmers(ref, k)
            synth.write_kmer_list(tmp + "/rep.txt", km, cnt, k)
            kf = (tmp + "/rep.txt").encode()
        h = H.h_index_build_flag(fa.encode(), kf, k, w, hpc, 4)
        mi = R.refshim_idx_build_flag(fa.encode(), kf, k, w, hpc, 4)
        opt = R.refshim_mapopt(preset.encode(), 0x4 | 0x20, mi)
        for i, r in enumerate(reads):
            s = synth.codes_to_ascii(r)
            ho = np.zeros(16 * 512, np.int32); co = np.zeros(8000000, np.uint32); nc = C.c_int64(); st = np.zeros(4, np.uint64)
            n = H.h_map(h, preset.encode(), 0x4 | 0x20, s, len(s), b"q", ho, 512, co, len(co), C.byref(nc), st.ctypes.data)
            rh = np.zeros(16 * 512, np.int32); rc = np.zeros(8000000, np.uint32); rnc = C.c_int64()
            rn = R.refshim_map(mi, opt, s, len(s), b"q", rh, 512, rc, len(rc), C.byref(rnc))
            a = ho[:16 * max(n, 0)].reshape(-1, 16).copy(); b = rh[:16 * max(rn, 0)].reshape(-1, 16).copy()
            if preset != "splice" and len(s) >= 10000 and not H.h_last_rep_len_defined():          # pure MCAS path: the reference's MAPQ comes from an uninitialised rep_len (src/map.c:281); with a rescan (:808-813) or the fallback (:859-861) it is assigned and compared
                a[:, 6] = 0; b[:, 6] = 0
            else:
                with_mapq += 1
            reads_total += 1
            if n != rn or not np.array_equal(a, b) or nc.value != rnc.value or not np.array_equal(co[:nc.value], rc[:rnc.value]):
                bad += 1
                if verbose:
                    print("MISMATCH seed", seed, "preset", preset, "read", i, "len", len(s), "hits", n, rn, flush=True)
        R.refshim_idx_destroy(mi)
    return reads_total, bad, with_mapq


if __name__ == "__main__":
    s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    t0 = time.time()
    n, bad, wq = run(s0, ns)
    print("host diff fuzz: seeds %d..%d, %d reads (%d with MAPQ compared), %d mismatching reads, %.0f s" % (s0, s0 + ns - 1, n, wq, bad, time.time() - t0), flush=True)
