"""Fuzz of the chunked sketch (sketch_p1_range / sketch_find_sync / sketch_p2_range, the sketch_long_* kernels' device functions) on the wavefront emulator
against the oracle's mm_sketch: random sequences (random bases, N runs, homopolymers, tandem repeats of period 1..40 with and without mutations, -W hits),
random odd k / w / chunk length.   python tools/sketch_chunk_fuzz.py [first_seed] [n_seeds]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import wmtest as W
from winnowmap_amd import build, synth


def main():
    s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    E = C.CDLL(build.build_emu())
    at = [C.c_int, W.u8p, W.u64p, W.i32p, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, W.u64p, W.u64p, W.u64p, W.i32p, W.i32p, C.c_int, C.POINTER(C.c_int32)]
    E.emu_sketch_chunked.argtypes = at
    n_seq = n_abs = 0
    for seed in range(s0, s0 + ns):
        rng = np.random.default_rng(seed)
        k = int(rng.choice([11, 13, 15, 17, 19, 21, 25]))
        w = int(rng.choice([5, 10, 19, 50, 50, 100]))
        chunk = int(rng.integers(64, 2000))
        seqs = []
        for it in range(12):
            L = int(rng.integers(1, 9000))
            kind = it % 4
            if kind == 0:
                s = rng.integers(0, 4, L).astype(np.uint8)
            elif kind == 1:
                unit = int(rng.integers(1, 41))
                s = np.tile(rng.integers(0, 4, unit), L // unit + 1)[:L].astype(np.uint8)
                if rng.random() < 0.5:
                    s = synth.mutate_codes(s, rng, 0.02, 0.0, 0.0)
            elif kind == 2:
                parts = []
                while sum(len(p) for p in parts) < L:
                    t = int(rng.integers(0, 3))
                    n = int(rng.integers(1, 700))
                    parts.append(rng.integers(0, 4, n).astype(np.uint8) if t == 0 else np.full(n, 4 if t == 1 else int(rng.integers(0, 4)), np.uint8))
                s = np.concatenate(parts)[:L]
            else:
                s = rng.integers(0, 4, L).astype(np.uint8)
                for _ in range(int(rng.integers(0, 6))):
                    s[int(rng.integers(0, L))] = 4
            seqs.append(np.ascontiguousarray(s))
        # a -W list made of k-mers of the sequences themselves, so that the down-weighted order (-x^8) takes part
        km = []
        for s in seqs[:4]:
            for p in range(0, max(0, len(s) - k), 37):
                kk = s[p:p + k]
                if len(kk) == k and (kk < 4).all():
                    f = r = 0
                    for c in kk:
                        f = (f << 2 | int(c)) & ((1 << 2 * k) - 1)
                    for c in kk[::-1]:
                        r = (r << 2 | (3 - int(c))) & ((1 << 2 * k) - 1)
                    km.append(min(f, r))
        bobj = W.o_bloom(km) if km and seed % 3 else None
        if bobj is None:
            tb, salts, bits = 8, (0, 0), np.zeros(8, np.uint8)
        else:
            tb, salts, bits = W.o_bloom_view(bobj)
        bloom = {"obj": bobj} if bobj is not None else None
        lens = np.array([len(s) for s in seqs], np.int32)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
        caps = (lens + 1).astype(np.int32)
        ooffs = np.concatenate([[0], np.cumsum(caps)[:-1]]).astype(np.uint64)
        ox = np.zeros(int(caps.sum()), np.uint64); oy = np.zeros(int(caps.sum()), np.uint64); counts = np.zeros(len(seqs), np.int32)
        ab = C.c_int32()
        E.emu_set_packed(seed & 1)                   # every other seed: the sequences as 2 bits per base + ambiguity bitmap (csrc/reads2bit.h)
        rc = E.emu_sketch_chunked(len(seqs), np.concatenate(seqs), offs, lens, w, k, tb, salts[0], salts[1], bits.ctypes.data, ox, oy, ooffs, caps, counts, chunk, C.byref(ab))
        E.emu_set_packed(0)
        assert rc == 0
        for i, s in enumerate(seqs):
            ex, ey = W.o_sketch(bytes(s), w, k, rid=0, bloom=bloom["obj"] if bloom else None)
            n = counts[i]
            ok = n == len(ex) and np.array_equal(ox[int(ooffs[i]):int(ooffs[i]) + n], ex) and np.array_equal(oy[int(ooffs[i]):int(ooffs[i]) + n], ey)
            if not ok:
                print("MISMATCH seed %d seq %d len %d w %d k %d chunk %d: %d vs %d minimizers" % (seed, i, len(s), w, k, chunk, n, len(ex)))
                sys.exit(1)
            n_seq += 1
        n_abs += ab.value
    print("%d seeds, %d sequences, %d absorbed chunks: 0 mismatches" % (ns, n_seq, n_abs))


if __name__ == "__main__":
    main()
