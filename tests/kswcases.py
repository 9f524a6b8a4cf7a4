"""Seeded ksw2 test cases shared by the emulator, oracle-vs-reference and GPU parity tests."""
import numpy as np
from winnowmap_amd import synth

FLAGS = [0x08, 0x00, 0x40, 0xC2, 0x42, 0x80]          # combinations used by src/align.c (+ two extras)
PRESETS = [(2, 4, 4, 2, 24, 1), (1, 4, 6, 2, 26, 1), (2, 4, 4, 2, 4, 2),   # (a,b,q,e,q2,e2): map-ont, asm20, single-affine,
           (1, 19, 39, 3, 81, 1), (1, 9, 16, 2, 41, 1)]                       # asm5, asm10 (src/options.c:104-111; large penalties: int8 wrap-around)


def make_cases(seed, n, max_len=700, preset=None):
    """Returns a list of dicts: q, t (uint8 codes), a, b, q_, e, q2, e2, w, zdrop, end_bonus, flag."""
    rng = np.random.default_rng(seed)
    out = []
    for it in range(n):
        tl = int(rng.integers(1, max_len)) if it % 3 else int(rng.integers(1, 60))
        t = rng.integers(0, 4, tl).astype(np.uint8)
        if it % 5 == 0:
            q = rng.integers(0, 4, int(rng.integers(1, max_len))).astype(np.uint8)
        else:
            q = synth.mutate_codes(t, rng, 0.03, 0.03, 0.04)
        if len(q) == 0:
            q = np.array([1], np.uint8)
        if it % 7 == 0:
            q[rng.integers(0, len(q))] = 4
        if it % 11 == 0:
            t[rng.integers(0, len(t))] = 4
        pr = PRESETS[it % len(PRESETS)] if preset is None else PRESETS[preset]
        out.append(dict(q=q, t=t, a=pr[0], b=pr[1], q_=pr[2], e=pr[3], q2=pr[4], e2=pr[5],
                        w=[751, 3001, 50, 10, 200, -1, 5][it % 7], zdrop=[400, 200, 25, 50, -1][it % 5],
                        end_bonus=[-1, 10, 0][it % 3], flag=FLAGS[it % 6]))
    return out


def ont_segments(seed, n, mean=300, w=751):
    """Gap-fill / extension shaped jobs like the map-ont workload (SURVEY.md §2 kernel table)."""
    rng = np.random.default_rng(seed)
    out = []
    for it in range(n):
        tl = int(np.clip(rng.lognormal(np.log(mean), 0.5), 60, 950))
        t = rng.integers(0, 4, tl).astype(np.uint8)
        q = synth.mutate_codes(t, rng, 0.03, 0.03, 0.04)
        if len(q) == 0:
            q = t.copy()
        r = it % 11
        flag = 0x40 if r == 0 else 0xC2 if r == 1 else 0x08
        out.append(dict(q=q, t=t, a=2, b=4, q_=4, e=2, q2=24, e2=1, w=w, zdrop=400, end_bonus=-1, flag=flag))
    return out


# ---- splice-aware extension (ksw_exts2_sse): a transcript-like query against a target with introns --------------------------------
SPLICE_FLAGS = (0x100, 0x200, 0x100 | 0x400, 0x200 | 0x400 | 0x80, 0x100 | 0x02, 0x100 | 0x08, 0x100 | 0x40, 0x200 | 0x40 | 0x80, 0x100 | 0x200, 0x00)


def make_splice_cases(seed, n, max_exon=120, max_intron=400):
    """exons joined in the query, separated by introns (canonical GT..AG, CT..AC for the reverse strand, or random) in the target; optional
    annotated junction bits; N bases; scoring = the splice preset (src/options.c:117-127) or a variant"""
    import numpy as np
    from winnowmap_amd import synth
    rng = np.random.default_rng(seed)
    out = []
    for it in range(n):
        n_exon = int(rng.integers(1, 5))
        q_parts, t_parts, junc = [], [], []
        t_len = 0
        flank = rng.integers(0, 4, int(rng.integers(0, 30))).astype(np.uint8)
        t_parts.append(flank); t_len += len(flank)
        for x in range(n_exon):
            ex = rng.integers(0, 4, int(rng.integers(8, max_exon))).astype(np.uint8)
            q_parts.append(synth.mutate_codes(ex, rng, 0.03, 0.01, 0.01) if it % 3 else ex)
            t_parts.append(ex); t_len += len(ex)
            if x + 1 < n_exon:
                intron = rng.integers(0, 4, int(rng.integers(20, max_intron))).astype(np.uint8)
                kind = int(rng.integers(0, 4))
                if kind == 0 and len(intron) >= 6:
                    intron[:3] = (2, 3, int(rng.integers(0, 4))); intron[-3:] = (int(rng.integers(0, 4)), 0, 2)      # GT. ... .AG
                elif kind == 1 and len(intron) >= 6:
                    intron[:3] = (1, 3, int(rng.integers(0, 4))); intron[-3:] = (int(rng.integers(0, 4)), 0, 1)      # CT. ... .AC
                junc.append((t_len, t_len + len(intron) - 1))
                t_parts.append(intron); t_len += len(intron)
        tail = rng.integers(0, 4, int(rng.integers(0, 30))).astype(np.uint8)
        t_parts.append(tail)
        q = np.concatenate(q_parts); t = np.concatenate(t_parts)
        if it % 7 == 3:
            t[int(rng.integers(0, len(t)))] = 4
        if it % 11 == 5:
            q[int(rng.integers(0, len(q)))] = 4
        flag = SPLICE_FLAGS[it % len(SPLICE_FLAGS)]
        if flag & 0x80:
            q = q[::-1].copy(); t = t[::-1].copy()
        jn = None
        if it % 2 == 0:
            jn = np.zeros(len(t), np.uint8)
            for (a, b) in junc:      # donor / acceptor bits on both strands (src/index.c:690-803: 1 | 8 at the intron start, 2 | 4 at its end)
                if flag & 0x80:
                    a, b = len(t) - 1 - b, len(t) - 1 - a
                jn[a] |= 1 | 8; jn[b] |= 2 | 4
        sc = [(1, 2, 2, 1, 32, 9, 9), (1, 2, 2, 1, 32, 9, 0), (2, 4, 4, 2, 40, 5, 3), (1, 1, 3, 1, 24, 7, 2)][it % 4]
        out.append(dict(q=q, t=t, a=sc[0], b=sc[1], q_=sc[2], e=sc[3], q2=sc[4], noncan=sc[5], junc_bonus=sc[6],
                        zdrop=[200, 50, -1, 400][it % 4], flag=flag, junc=jn))
    return out


def stripe_cases(seed, n, max_len):
    """approximate-maximum, exact and z-dropping jobs of any length (make_cases only has tiny approximate-maximum jobs): related prefix + unrelated
    tail, long indels, N bases, every band"""
    from winnowmap_amd import synth
    rng = np.random.default_rng(seed)
    out = []
    for ci in range(n):
        tl = int(rng.integers(1, max_len))
        t = rng.integers(0, 4, tl).astype(np.uint8)
        kind = int(rng.integers(0, 6))
        if kind == 0:
            q = rng.integers(0, 4, int(rng.integers(1, max_len))).astype(np.uint8)
        elif kind == 1:
            k = int(rng.integers(1, tl + 1))
            q = np.concatenate([synth.mutate_codes(t[:k], rng, 0.03, 0.03, 0.04), rng.integers(0, 4, int(rng.integers(1, max_len))).astype(np.uint8)])
        elif kind == 2:
            k = int(rng.integers(0, tl + 1)); g = int(rng.integers(1, 200))
            q = synth.mutate_codes(t, rng, 0.02, 0.02, 0.02)
            q = np.concatenate([q[:k], rng.integers(0, 4, g).astype(np.uint8), q[k:]]) if rng.integers(0, 2) else np.concatenate([q[:k], q[min(len(q), k + g):]])
        else:
            q = synth.mutate_codes(t, rng, 0.03, 0.03, 0.04)
        if len(q) == 0:
            q = np.array([1], np.uint8)
        if rng.integers(0, 8) == 0:
            q[rng.integers(0, len(q))] = 4
        if rng.integers(0, 8) == 0:
            t[rng.integers(0, len(t))] = 4
        out.append(dict(q=q, t=t, a=2, b=4, q_=4, e=2, q2=24, e2=1,
                        w=int(rng.choice([751, 3001, 50, 10, 200, -1, 5, 100, 127, 128, 129, 30, 1501])), zdrop=int(rng.choice([400, 200, 25, 50, -1, 100])),
                        end_bonus=int(rng.choice([-1, 10, 0])), flag=int(rng.choice([0x08, 0x08, 0x00, 0x40, 0xC2, 0x42, 0x80, 0x0A, 0x88]))))
    return out


def stripe_edge_cases(seed, n, max_len):
    """lengths around the stripe boundaries (multiples of 128), unequal lengths, every small band: the ends of a job as the stripe-pipelined kernel sees
    them (a band that runs empty before the next stripe starts, stripes that are never reached, the band's last lane right below a stripe)"""
    from winnowmap_amd import synth
    rng = np.random.default_rng(seed)
    out = []
    for ci in range(n):
        base = int(rng.choice([0, 128, 256, 384, 512, 640])) if rng.integers(0, 2) else 0
        tl = max(1, min(max_len, base + int(rng.integers(-20, 140))))
        ql = max(1, int(rng.choice([tl + int(rng.integers(-150, 150)), int(rng.integers(1, max_len)), tl])))
        t = rng.integers(0, 4, tl).astype(np.uint8)
        if rng.integers(0, 3) == 0:
            q = rng.integers(0, 4, ql).astype(np.uint8)
        else:
            q = synth.mutate_codes(t, rng, 0.03, 0.03, 0.04)
            q = np.concatenate([q, rng.integers(0, 4, max(0, ql - len(q))).astype(np.uint8)])[:ql]
        if len(q) == 0:
            q = np.array([1], np.uint8)
        if rng.integers(0, 10) == 0:
            q[rng.integers(0, len(q))] = 4
        if rng.integers(0, 10) == 0:
            t[rng.integers(0, len(t))] = 4
        pr = PRESETS[int(rng.integers(0, len(PRESETS)))]
        out.append(dict(q=q, t=t, a=pr[0], b=pr[1], q_=pr[2], e=pr[3], q2=pr[4], e2=pr[5],
                        w=int(rng.choice([1, 2, 5, 10, 15, 16, 17, 30, 31, 32, 33, 50, 63, 64, 65, 100, 111, 112, 113, 127, 128, 200, 751, -1])),
                        zdrop=int(rng.choice([400, 200, 25, 50, -1, 100])), end_bonus=int(rng.choice([-1, 10, 0])),
                        flag=int(rng.choice([0x08, 0x08, 0x00, 0x40, 0xC2, 0x42, 0x80, 0x0A, 0x88]))))
    return out


def dual_pairs(seed, n, ncs=(4, 8, 16)):
    """(nc, scoring, [job] or [job, job]) for the two-alignments-per-wavefront kernel (ksw_dual_kernel.h): gap fills — band never clips, no N, approximate
    maximum — of every size up to the window of nc 64-lane chunks: unrelated operands, related ones with indels, truncated queries, tiny targets, pairs of very
    different lengths (one alignment finishes long before the other), single jobs (the odd one of a launch), mixed KSW_EZ_RIGHT / REV_CIGAR / EXTZ_ONLY flags"""
    from winnowmap_amd import synth
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        nc = int(ncs[int(rng.integers(0, len(ncs)))])
        lim = 64 * nc - 16
        sc = dict(a=int(rng.integers(1, 4)), b=int(rng.integers(2, 7)), q_=int(rng.integers(2, 8)), e=int(rng.integers(1, 4)), q2=int(rng.integers(8, 30)), e2=1)
        jobs = []
        for x in range(1 if rng.random() < 0.15 else 2):
            style = int(rng.integers(0, 4))
            tlen = int(rng.integers(1, lim)) if style else int(rng.integers(1, 40))
            t = rng.integers(0, 4, tlen).astype(np.uint8)
            if style == 1:
                q = rng.integers(0, 4, int(rng.integers(1, lim))).astype(np.uint8)
            else:
                q = synth.mutate_codes(t, rng, 0.05, 0.04, 0.04)
                if len(q) == 0:
                    q = t[:1].copy()
                if rng.random() < 0.3:
                    q = q[:max(1, int(rng.integers(1, len(q) + 1)))]
            q = np.ascontiguousarray(q[:2000], np.uint8)
            if min(len(q), tlen) > lim - 1:
                continue
            flag = 0x08 | (0x02 if rng.random() < 0.5 else 0) | (0x80 if rng.random() < 0.3 else 0) | (0x40 if rng.random() < 0.1 else 0)
            jobs.append(dict(q=q, t=t, w=-1 if rng.random() < 0.5 else max(len(q), tlen) + int(rng.integers(0, 50)), flag=flag, zdrop=-1, end_bonus=0, **sc))
        if jobs:
            out.append((nc, sc, jobs))
    return out
