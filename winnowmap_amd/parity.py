"""Text-level parity of PAF / SAM records against the reference binary's output (bench.py's `parity` block and the
at-scale GPU tests). Compared: every column and tag except MAPQ and `rl:i` (the reference computes both from an
uninitialised `rep_len` on the MCAS stage-2 path, /root/reference/src/map.c:281, so it does not reproduce them itself),
the MAPQ field inside SA:Z, and the @PG header line. Records are grouped by read name, so the two texts may list the
reads in different orders (the reference prints a mini-batch longest read first, src/map.c:1124-1143); inside a read
the order of the records must agree."""


def mask_record(line, sam):
    """One output line with the non-reproducible fields removed; None for header lines that are not compared."""
    if sam and line.startswith(b"@"):
        return None
    f = line.rstrip(b"\n").split(b"\t")
    if sam:
        if len(f) > 4:
            f[4] = b"*"
    elif len(f) > 11:
        f[11] = b"*"
    out = []
    for x in f:
        if x.startswith(b"rl:i:"):
            continue
        if sam and x.startswith(b"SA:Z:"):
            x = b"SA:Z:" + b";".join(b",".join(p.split(b",")[:4] + [b"*"] + p.split(b",")[5:]) for p in x[5:].split(b";") if p)
        out.append(x)
    return b"\t".join(out)


def group_by_read(text, sam=False):
    g = {}
    for line in text.split(b"\n"):
        if not line:
            continue
        m = mask_record(line, sam)
        if m is None:
            continue
        name = m[:m.index(b"\t")] if b"\t" in m else m
        g.setdefault(name, []).append(m)
    return g


def diff_texts(ref_text, our_text, sam=False, max_examples=3):
    """-> dict(reads, hits, mismatches, cigar_ops, examples): `mismatches` counts reads whose record lists differ in any way."""
    a = group_by_read(ref_text, sam)
    b = group_by_read(our_text, sam)
    mism = 0
    hits = 0
    examples = []
    for name in a.keys() | b.keys():
        ra, rb = a.get(name, []), b.get(name, [])
        hits += len(ra)
        if ra != rb:
            mism += 1
            if len(examples) < max_examples:
                k = next((i for i in range(min(len(ra), len(rb))) if ra[i] != rb[i]), min(len(ra), len(rb)))
                examples.append({"read": name.decode(errors="replace"), "n_ref": len(ra), "n_ours": len(rb),
                                 "ref": (ra[k][:300].decode(errors="replace") if k < len(ra) else None),
                                 "ours": (rb[k][:300].decode(errors="replace") if k < len(rb) else None)})
    return {"reads": len(a.keys() | b.keys()), "hits": hits, "mismatches": mism, "examples": examples}
