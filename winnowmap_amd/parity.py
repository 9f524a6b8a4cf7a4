"""Text-level parity of PAF / SAM records against the reference binary's output (bench.py's `parity` block and the
at-scale GPU tests). Compared: every column and tag. MAPQ, `rl:i` and the MAPQ field inside SA:Z are masked ONLY for reads the
reference maps through its two-stage MCAS procedure — reads of at least `mcas_gate` bases (mm_mapopt_t::SVawareMinReadLength,
10 000 by default) outside splice mode: there `mm_set_mapq` (/root/reference/src/hit.c:463-508) is fed an uninitialised `rep_len`
(src/map.c:281, never assigned on that path, read at :933), so the reference does not reproduce the two fields itself. Below the
gate (src/map.c:859-861 sets rep_len) and in splice mode every field is compared. Round 6: ABOVE the gate the reference still assigns rep_len
for the reads whose stage-1 pass left stretches unmapped (the rescan, src/map.c:808-813) or found nothing (the fallback, :859-861); the mapper
reports those reads (wm_map_reads_rep_len_defined) and `defined` = their names makes MAPQ / rl:i part of the comparison for them too.
The @PG header line is never compared. Records
are grouped by read name, so the two texts may list the reads in different orders (the reference prints a mini-batch longest read
first, src/map.c:1124-1143); inside a read the order of the records must agree."""

MCAS_GATE = 10000


def sam_query_len(cigar):
    """length of the whole read from a SAM CIGAR (M I S H = X consume the read; hard clips count: the read is longer than SEQ)"""
    n = tot = 0
    for ch in cigar:
        if 48 <= ch <= 57:
            n = n * 10 + ch - 48
        else:
            if ch in b"MISH=X":
                tot += n
            n = 0
    return tot


def record_query_len(f, sam):
    """read length of a split record (None if it cannot be told: an unmapped SAM record gives len(SEQ))"""
    try:
        if not sam:
            return int(f[1])
        if f[5] == b"*":
            return len(f[9]) if f[9] != b"*" else None
        return sam_query_len(f[5])
    except (IndexError, ValueError):
        return None


def mask_record(line, sam, mcas_gate=MCAS_GATE, defined=None):
    """One output line with the non-reproducible fields removed; None for header lines that are not compared.
    mcas_gate: reads at least this long have MAPQ / rl:i masked; None: nothing is masked (splice mode, or MCAS switched off).
    defined: names (bytes) of reads at or above the gate whose rep_len the reference assigns: not masked either."""
    if sam and line.startswith(b"@"):
        return None
    f = line.rstrip(b"\n").split(b"\t")
    qlen = record_query_len(f, sam)
    if mcas_gate is None or (qlen is not None and qlen < mcas_gate) or (defined is not None and f[0] in defined):
        return b"\t".join(f)
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


def group_by_read(text, sam=False, mcas_gate=MCAS_GATE, defined=None):
    g = {}
    for line in text.split(b"\n"):
        if not line:
            continue
        m = mask_record(line, sam, mcas_gate, defined)
        if m is None:
            continue
        name = m[:m.index(b"\t")] if b"\t" in m else m
        g.setdefault(name, []).append(m)
    return g


def defined_names(names, flags):
    """the `defined` argument of diff_texts from the read names of a Mapper.map() call and Mapper.rep_len_defined()"""
    return {(n if isinstance(n, bytes) else n.encode()) for n, f in zip(names, flags) if f}


def diff_texts(ref_text, our_text, sam=False, max_examples=3, mcas_gate=MCAS_GATE, defined=None):
    """-> dict(reads, hits, mismatches, mapq_compared, examples): `mismatches` counts reads whose record lists differ in any way;
    `mapq_compared` = records of the reference whose MAPQ (and rl:i) took part in the comparison."""
    a = group_by_read(ref_text, sam, mcas_gate, defined)
    b = group_by_read(our_text, sam, mcas_gate, defined)
    mism = 0
    hits = 0
    with_mapq = 0
    col = 4 if sam else 11
    examples = []
    for name in a.keys() | b.keys():
        ra, rb = a.get(name, []), b.get(name, [])
        hits += len(ra)
        for rec in ra:
            f = rec.split(b"\t")
            with_mapq += len(f) > col and f[col] != b"*"
        if ra != rb:
            mism += 1
            if len(examples) < max_examples:
                k = next((i for i in range(min(len(ra), len(rb))) if ra[i] != rb[i]), min(len(ra), len(rb)))
                examples.append({"read": name.decode(errors="replace"), "n_ref": len(ra), "n_ours": len(rb),
                                 "ref": (ra[k][:300].decode(errors="replace") if k < len(ra) else None),
                                 "ours": (rb[k][:300].decode(errors="replace") if k < len(rb) else None)})
    return {"reads": len(a.keys() | b.keys()), "hits": hits, "mismatches": mism, "mapq_compared": with_mapq, "examples": examples}
