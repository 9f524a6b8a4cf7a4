"""Seeded synthetic inputs for the parity tests and bench.py: references with implanted
repeat families, ONT / HiFi-like reads of EXACT length (so the L >= 10000 MCAS gate of the reference,
src/map.c:314, is deterministic), and the `-W` list (canonical k-mers above the 0.9998-distinct count
threshold, README.md:29-30 / ext/meryl/src/meryl/merylOp-nextMer.C:103-115) that meryl would produce.

Pure numpy; no reference code involved."""
import numpy as np

_ACGT = np.frombuffer(b"ACGTN", np.uint8)          # code 4 = ambiguous base
_COMP = np.array([3, 2, 1, 0, 4], np.uint8)


def random_codes(n, rng):
    return rng.integers(0, 4, n, dtype=np.uint8)


def mutate_codes(codes, rng, sub, ins, dele):
    """i.i.d. substitution / insertion / deletion channel on 0..3 codes."""
    n = len(codes)
    r = rng.random(n)
    keep = r >= dele
    is_sub = (r >= dele) & (r < dele + sub)
    out = codes.copy()
    out[is_sub] = (out[is_sub] + rng.integers(1, 4, int(is_sub.sum()), dtype=np.uint8)) & 3
    ins_mask = keep & (rng.random(n) < ins)
    # build output with insertions after kept bases
    cnt = keep.astype(np.int64) + ins_mask.astype(np.int64)
    pos = np.cumsum(cnt) - cnt
    res = np.empty(int(cnt.sum()), np.uint8)
    res[pos[keep]] = out[keep]
    res[pos[ins_mask] + 1] = rng.integers(0, 4, int(ins_mask.sum()), dtype=np.uint8)
    return res


def make_reference(n_contigs, contig_len, seed, repeat_frac=0.0, satellite=True):
    """List of uint8 code arrays. With repeat_frac > 0 implants: copies of a 5 kb
    element at 2 % divergence and, if `satellite`, tandem arrays of a 171-bp monomer at 2 % per-copy divergence."""
    rng = np.random.default_rng(seed)
    contigs = [random_codes(contig_len, rng) for _ in range(n_contigs)]
    total = n_contigs * contig_len
    if repeat_frac > 0:
        elem = random_codes(5000, rng)
        n_elem = max(1, int(total * repeat_frac * 0.5 / 5000))
        for _ in range(n_elem):
            c = contigs[int(rng.integers(0, n_contigs))]
            cp = mutate_codes(elem, rng, 0.02, 0.0, 0.0)
            p = int(rng.integers(0, max(1, len(c) - len(cp))))
            c[p:p + len(cp)] = cp[:len(c) - p]
        if satellite:
            mono = random_codes(171, rng)
            arr_len = max(171 * 4, int(total * repeat_frac * 0.5 / 3))
            for _ in range(3):
                c = contigs[int(rng.integers(0, n_contigs))]
                arr_len_c = min(arr_len, len(c) // 2)
                ncopy = arr_len_c // 171
                arr = np.concatenate([mutate_codes(mono, rng, 0.02, 0.0, 0.0) for _ in range(ncopy)])
                p = int(rng.integers(0, max(1, len(c) - len(arr))))
                c[p:p + len(arr)] = arr[:len(c) - p]
    return contigs


def revcomp_codes(c):
    return _COMP[c[::-1]]


PROFILES = {"ont": (0.03, 0.03, 0.04), "hifi": (0.001, 0.0005, 0.0005), "exact": (0.0, 0.0, 0.0)}


def make_reads(contigs, n_reads, read_len, seed, profile="ont", sv_frac=0.0):
    """Reads of exactly read_len codes, uniform start, random strand. Returns (list of code arrays, truth list)."""
    rng = np.random.default_rng(seed)
    sub, ins, dele = PROFILES[profile]
    lens = np.array([len(c) for c in contigs])
    reads, truth = [], []
    span = int(read_len * (1.0 + 1.5 * dele + 0.02)) + 64
    for i in range(n_reads):
        while True:
            ci = int(rng.integers(0, len(contigs)))
            if lens[ci] > span + 3000:
                break
        st = int(rng.integers(0, lens[ci] - span - 3000))
        src = contigs[ci][st:st + span + 3000]
        if sv_frac > 0 and rng.random() < sv_frac:
            kind = int(rng.integers(0, 3))
            mid = len(src) // 2
            if kind == 0:
                src = np.concatenate([src[:mid], random_codes(2000, rng), src[mid:]])
            elif kind == 1:
                src = np.concatenate([src[:mid], src[mid + 1000:]])
            else:
                src = np.concatenate([src[:mid], revcomp_codes(src[mid:mid + 500]), src[mid + 500:]])
        r = mutate_codes(src, rng, sub, ins, dele)
        while len(r) < read_len:  # cannot happen with the margins above, but keep the length exact
            r = np.concatenate([r, random_codes(read_len - len(r), rng)])
        r = r[:read_len]
        strand = int(rng.integers(0, 2))
        if strand:
            r = revcomp_codes(r)
        reads.append(np.ascontiguousarray(r))
        truth.append((ci, st, strand))
    return reads, truth


def codes_to_ascii(c):
    return _ACGT[c].tobytes()


def write_fasta(path, seqs, prefix="s"):
    with open(path, "wb") as f:
        for i, s in enumerate(seqs):
            f.write(b">%s%d\n" % (prefix.encode(), i))
            f.write(codes_to_ascii(s) if s.dtype == np.uint8 and s.max(initial=0) < 5 else bytes(s))
            f.write(b"\n")


def canonical_kmers(codes, k):
    """Canonical 2-bit k-mers of one code array (uint64), min(forward, revcomp) as src/index.c:362-376."""
    n = len(codes) - k + 1
    if n <= 0:
        return np.zeros(0, np.uint64)
    c = codes.astype(np.uint64)
    fw = np.zeros(n, np.uint64)
    rc = np.zeros(n, np.uint64)
    for j in range(k):
        fw = (fw << np.uint64(2)) | c[j:j + n]
        rc = rc | ((np.uint64(3) - c[j:j + n]) << np.uint64(2 * j))
    return np.minimum(fw, rc)


def repetitive_kmers(contigs, k, distinct=0.9998):
    """The `-W` list: canonical k-mers whose count exceeds the smallest threshold c for which at least
    `distinct` of the distinct k-mers have count <= c (meryl `print greater-than distinct=0.9998`).
    Returns (kmers uint64 sorted, counts)."""
    km = np.concatenate([canonical_kmers(c, k) for c in contigs])
    uniq, cnt = np.unique(km, return_counts=True)
    if len(uniq) == 0:
        return uniq, cnt
    # merylOp-nextMer.C:103-115: integer (truncated) target, walk the count values that occur, first cumulative count >= target
    vals, occ = np.unique(cnt, return_counts=True)
    cum = np.cumsum(occ)
    target = int(distinct * len(uniq))
    thr = int(vals[int(np.searchsorted(cum, target, side="left"))])
    sel = cnt > thr
    return uniq[sel], cnt[sel]


def kmer_to_str(x, k):
    return "".join("ACGT"[(int(x) >> (2 * (k - 1 - i))) & 3] for i in range(k))


def write_kmer_list(path, kmers, counts, k):
    with open(path, "w") as f:
        for x, c in zip(kmers, counts):
            f.write("%s\t%d\n" % (kmer_to_str(x, k), int(c)))


def make_transcripts(ref, n, seed, sub=0.02, ins=0.01, dele=0.01, canonical=0.8):
    """n spliced reads: 2..5 exons of 80..300 bases separated by introns of 150..3000 bases, taken from `ref` (list of code arrays, MODIFIED in
    place: GT..AG is planted at a fraction `canonical` of the introns), every second read from the reverse strand. For splice-mode tests."""
    rng = np.random.default_rng(seed)
    reads = []
    for g in range(n):
        seq = ref[int(rng.integers(0, len(ref)))]
        pos = int(rng.integers(1000, len(seq) - 30000))
        parts = []
        for _ in range(int(rng.integers(2, 6))):
            el = int(rng.integers(80, 300))
            parts.append(seq[pos:pos + el].copy())
            pos += el
            il = int(rng.integers(150, 3000))
            if rng.random() < canonical:
                seq[pos:pos + 2] = (2, 3)
                seq[pos + il - 2:pos + il] = (0, 2)
            pos += il
        tr = mutate_codes(np.concatenate(parts), rng, sub, ins, dele)
        reads.append(revcomp_codes(tr) if g % 2 else tr)
    return reads
