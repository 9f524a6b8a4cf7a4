"""Fuzz of the homopolymer-compressed sketch (MM_I_HPC, src/sketch.c:152-163) on the wavefront emulator: sketch_coop with wm_sketch_params_t::hpc over random
sequences with runs of every length (beyond the 255-base span limit too), N next to runs, short sequences; bytes and packed reads, random odd k and w, with and
without a -W filter — against the oracle's HPC branch (pinned to the reference's mm_sketch, tests/test_oracle_vs_ref.py).
   python tools/sketch_hpc_fuzz.py [first_seed] [n_seeds]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import wmtest as W
from winnowmap_amd import build


def main():
    s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    E = C.CDLL(build.build_emu())
    E.emu_sketch_coop.argtypes = [C.c_int, W.u8p, W.u64p, W.i32p, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, W.u64p, W.u64p, W.u64p, W.i32p, W.i32p]
    n_seq = n_span = 0
    for seed in range(s0, s0 + ns):
        rng = np.random.default_rng(seed)
        k = int(rng.choice([7, 11, 15, 19, 21, 27])); w = int(rng.choice([5, 10, 19, 50, 100]))
        seqs = []
        for it in range(10):
            L = int(rng.integers(1, 1200))
            runs = rng.integers(1, int(rng.choice([2, 4, 9])), L)
            if L > 20:
                runs[rng.integers(0, L, max(1, L // 30))] = rng.integers(10, 420, max(1, L // 30))
            base = rng.integers(0, 4, L)
            base[1:] = np.where(base[1:] == base[:-1], (base[1:] + 1) & 3, base[1:])
            s = np.repeat(base, runs).astype(np.uint8)
            for _ in range(int(rng.integers(0, 4))):
                p = int(rng.integers(0, len(s)))
                s[p:p + int(rng.integers(1, 5))] = 4
            seqs.append(s)
        bloom = None
        tb, salts, bits = 8, (0, 0), np.zeros(8, np.uint8)
        if seed % 2:                           # with a -W filter
            # (the filter's content does not matter for parity, only that both sides probe the same table: random k-mers)
            km = rng.integers(0, 1 << (2 * k), 300, dtype=np.uint64)
            bloom = W.o_bloom(int(v) for v in km)
            tb, salts, bits = W.o_bloom_view(bloom)
        lens = np.array([len(s) for s in seqs], np.int32)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
        caps = (lens + 1).astype(np.int32)
        ooffs = np.concatenate([[0], np.cumsum(caps)[:-1]]).astype(np.uint64)
        ox = np.zeros(int(caps.sum()), np.uint64); oy = np.zeros(int(caps.sum()), np.uint64); counts = np.zeros(len(seqs), np.int32)
        E.emu_set_packed(seed & 1 ^ (seed >> 1 & 1)); E.emu_set_hpc(1)
        E.emu_sketch_coop(len(seqs), np.concatenate(seqs), offs, lens, w, k, tb, salts[0], salts[1], bits.ctypes.data, ox, oy, ooffs, caps, counts)
        E.emu_set_packed(0); E.emu_set_hpc(0)
        for i, s in enumerate(seqs):
            ex, ey = W.o_sketch(bytes(s), w, k, rid=0, bloom=bloom, hpc=True)
            n = counts[i]
            if n != len(ex) or not np.array_equal(ox[int(ooffs[i]):int(ooffs[i]) + n], ex) or not np.array_equal(oy[int(ooffs[i]):int(ooffs[i]) + n], ey):
                print("MISMATCH seed", seed, "seq", i, "len", len(s), "k", k, "w", w, flush=True)
                sys.exit(1)
            n_seq += 1; n_span += int(np.count_nonzero((ex & np.uint64(0xff)) != np.uint64(k)))
    print("%d seeds, %d sequences, %d minimizers with a span other than k: 0 mismatches" % (ns, n_seq, n_span), flush=True)


if __name__ == "__main__":
    main()
