"""TEST INFRASTRUCTURE: the reference's own splice-mode alignment jobs (inputs AND results), replayed through the oracle and the kernel emulator.
  WM_SUBST=off WM_DUMP_EXTS2=/tmp/jobs.bin oracle/_ref/winnowmap_subst -t 2 -cx splice ref.fa reads.fa > /dev/null
  python tests/host_harness/replay_exts2_jobs.py /tmp/jobs.bin
Each record = one ksw_exts2_sse call of the untouched reference (oracle/wm_subst.cpp dumps it) with the ksw_extz_t it returned."""
import sys, os, struct, ctypes as C, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import wmtest as W
from winnowmap_amd import build
emu = C.CDLL(build.build_emu())
emu.emu_ksw_exts2.argtypes = [C.c_int, W.u8p, C.c_int, W.u8p, W.i8p] + [C.c_int] * 7 + [C.c_void_p, W.i32p, W.u32p, C.c_int]
data = open(sys.argv[1], "rb").read(); pos = 0
n = bad_o = bad_e = skipped = 0
flags = collections.Counter(); introns = 0
FIELDS = ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score")
while pos < len(data):
    h = struct.unpack_from("24i", data, pos); pos += 96
    ql, tl, m, m0, m1, m24, q, e, q2, noncan, zdrop, jb, flag, has_j = h[:14]
    want = dict(zip(FIELDS, h[14:23])); nc = h[23]
    qs = np.frombuffer(data, np.uint8, max(ql, 0), pos).copy(); pos += max(ql, 0)
    ts = np.frombuffer(data, np.uint8, max(tl, 0), pos).copy(); pos += max(tl, 0)
    jn = None
    if has_j and tl > 0:
        jn = np.frombuffer(data, np.uint8, tl, pos).copy(); pos += tl
    cg = np.frombuffer(data, np.uint32, max(nc, 0), pos).copy(); pos += 4 * max(nc, 0)
    if ql <= 0 or tl <= 0 or m != 5 or (flag & (0x01 | 0x04 | 0x10)):
        skipped += 1; continue
    n += 1; flags[flag] += 1; introns += any((int(x) & 0xf) == 3 for x in cg)
    mat = W.simple_mat(m0, -m1, -m24)
    o = W.o_ksw_exts2(qs, ts, mat=mat, q=q, e=e, q2=q2, noncan=noncan, zdrop=zdrop, junc_bonus=jb, flag=flag, junc=jn)
    if any(o[k] != want[k] for k in FIELDS) or not np.array_equal(o["cigar"], cg):
        bad_o += 1; print("ORACLE != REFERENCE", n, ql, tl, hex(flag), {k: (o[k], want[k]) for k in FIELDS if o[k] != want[k]})
    ez = np.zeros(10, np.int32); cig = np.zeros(ql + tl + 4, np.uint32)
    k = emu.emu_ksw_exts2(ql, qs, tl, ts, mat, q, e, q2, noncan, zdrop, jb, flag, None if jn is None else jn.ctypes.data, ez, cig, len(cig))
    if k < 0 or any(int(ez[i]) != want[f] for i, f in enumerate(FIELDS)) or not np.array_equal(cig[:max(k, 0)], cg):
        bad_e += 1; print("EMULATED KERNEL != REFERENCE", n, ql, tl, hex(flag))
print(n, "jobs (", skipped, "skipped: score-only / degenerate );", introns, "CIGARs with introns; flags", {hex(k): v for k, v in sorted(flags.items())})
print("oracle vs reference:", bad_o, "mismatches; emulated kernel vs reference:", bad_e, "mismatches")
