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
