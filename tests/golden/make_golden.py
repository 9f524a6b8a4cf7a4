"""Generates the end-to-end golden fixtures by running the REAL reference (oracle/_ref, built from /root/reference)
on seeded synthetic inputs. Run in the build container:  python tests/golden/make_golden.py
The fixtures hold, per read, the reference's hits (rid rs re qs qe rev . n_cigar score cnt mlen blen dp_score dp_max
dp_max2 flags; MAPQ kept for reads below the 10 kb MCAS gate and zeroed above it, where the reference does not reproduce it itself:
uninitialised rep_len, src/map.c:281) and the CIGARs.
The inputs are regenerated from the same seeds by the tests (winnowmap_amd/synth.py)."""
import ctypes as C
import os
import sys
import tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import wmtest as W  # noqa: E402
from winnowmap_amd import synth  # noqa: E402

CASES = {
    # name: (preset, ref seed, n contigs, contig len, repeat frac, reads seed, n reads, read len, profile, sv frac, use -W)
    "ont": ("map-ont", 31, 2, 400000, 0.1, 32, 24, 15000, "ont", 0.25, True),
    "ont_short": ("map-ont", 33, 1, 300000, 0.0, 34, 16, 6000, "ont", 0.0, False),
    "hifi": ("map-pb", 35, 2, 300000, 0.1, 36, 10, 20000, "hifi", 0.2, True),
    "asm20": ("asm20", 37, 1, 600000, 0.05, 38, 3, 80000, "hifi", 0.3, False),
    "ont_hpc": ("map-ont", 41, 1, 300000, 0.05, 42, 14, 9000, "ont", 0.15, True),      # -H: homopolymer-compressed index (IDX_FLAG), runs planted in the reference
}
IDX_FLAG = {"ont_hpc": 1}      # mm_idxopt_t::flag of the case's index (MM_I_HPC = 1, the CLI's -H); 0 where absent


def inputs(name, tmpdir):
    preset, rs, nc, cl, rf, qs, nr, rl, prof, sv, use_w = CASES[name]
    ref = synth.make_reference(nc, cl, rs, repeat_frac=rf)
    if IDX_FLAG.get(name, 0) & 1:               # homopolymer runs of 4 .. 280 bases every 2.5 kb: what -H is about
        rng = np.random.default_rng(rs + 1000)
        for c in ref:
            for p in range(1200, len(c) - 1500, 2500):
                c[p:p + int(rng.choice([4, 8, 17, 40, 280]))] = int(rng.integers(0, 4))
    fa = os.path.join(tmpdir, name + ".fa")
    synth.write_fasta(fa, ref)
    k = 19 if preset.startswith("asm") else 15
    kf = None
    if use_w:
        km, cnt = synth.repetitive_kmers(ref, k)
        kf = os.path.join(tmpdir, name + ".rep.txt")
        synth.write_kmer_list(kf, km, cnt, k)
    reads, _ = synth.make_reads(ref, nr, rl, qs, profile=prof, sv_frac=sv)
    return preset, fa, kf, k, [synth.codes_to_ascii(r) for r in reads]


def main():
    R = W.ref()
    R.refshim_idx_build_flag.restype = C.c_void_p
    R.refshim_idx_build_flag.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
    tmp = tempfile.mkdtemp()
    for name in (sys.argv[1:] or CASES):          # (python make_golden.py [case ...]: only the named fixtures are rewritten)
        preset, fa, kf, k, reads = inputs(name, tmp)
        mi = R.refshim_idx_build_flag(fa.encode(), (kf or "").encode(), k, 50, IDX_FLAG.get(name, 0), 4)
        opt = R.refshim_mapopt(preset.encode(), 0x4 | 0x20, mi)
        hits, cigs, first = [], [], [0]
        for i, s in enumerate(reads):
            h = np.zeros(16 * 256, np.int32)
            c = np.zeros(2000000, np.uint32)
            nc = C.c_int64()
            n = R.refshim_map(mi, opt, s, len(s), ("read%d" % i).encode(), h, 256, c, len(c), C.byref(nc))
            hh = h[:16 * n].reshape(-1, 16).copy()
            if len(s) >= 10000:
                hh[:, 6] = 0
            hits.append(hh)
            cigs.append(c[:nc.value].copy())
            first.append(first[-1] + n)
        np.savez_compressed(os.path.join(HERE, "e2e_%s.npz" % name), hits=np.concatenate(hits), cigars=np.concatenate(cigs), first=np.array(first, np.int64))
        print(name, "reads", len(reads), "hits", first[-1], "cigar ops", sum(len(x) for x in cigs))


if __name__ == "__main__":
    main()
