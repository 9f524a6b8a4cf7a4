"""TEST INFRASTRUCTURE: replays the alignment jobs of a real mapping run through the kernel emulator.
  WM_DUMP_KSW=/tmp/jobs.bin tests/host_harness/prof_main ref.fa rep.txt reads.fa 40 map-ont      (harness + oracle ops: dumps every ksw job)
  python tests/host_harness/replay_ksw_jobs.py /tmp/jobs.bin default,ror [max_jobs]
Every job goes through emu_ksw_extd2 with the class the product host would choose and is compared with the oracle (all ksw_extz_t fields and
the CIGAR). Complements the synthetic fuzz of tests/test_kernels_emu.py with the real distribution of flags / bands / z-drops / lengths."""
import sys, struct, ctypes as C, numpy as np, collections, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import wmtest as W
from winnowmap_amd import build
path = sys.argv[1]; variants = sys.argv[2].split(","); limit = int(sys.argv[3]) if len(sys.argv) > 3 else 10**9; force = int(sys.argv[4]) if len(sys.argv) > 4 else -1      # force: a class code of emu_ksw_extd2 (e.g. 223 = ksw_dp_pmulti<4,4>, CLIP + HASN)
libs = {}
for v in variants:
    E = C.CDLL(build.build_emu(() if v == "default" else ("WM_KSW_ROR=0",)))
    E.emu_ksw_extd2.argtypes = [C.c_int, W.u8p, C.c_int, W.u8p, W.i8p] + [C.c_int] * 9 + [W.i32p, W.u32p, C.c_int, C.POINTER(C.c_int)]
    libs[v] = E
data = open(path, 'rb').read(); pos = 0; n = 0; bad = 0; cells = 0
klass = collections.Counter(); t0 = time.time()
while pos < len(data) and n < limit:
    hdr = struct.unpack_from("10i", data, pos); pos += 40
    ql, tl, w, zdrop, end_bonus, flag, a, b, packed, amb = hdr
    q = np.frombuffer(data, np.uint8, ql, pos).copy(); pos += ql
    t = np.frombuffer(data, np.uint8, tl, pos).copy(); pos += tl
    qq, e, q2, e2 = packed & 0xff, packed >> 8 & 0xff, packed >> 16 & 0xff, packed >> 24 & 0xff
    if ql <= 0 or tl <= 0: continue
    mat = W.simple_mat(a, -b, -amb)
    o = W.o_ksw_extd2(q, t, mat=mat, q=qq, e=e, q2=q2, e2=e2, w=w, zdrop=zdrop, end_bonus=end_bonus, flag=flag)
    for v, E in libs.items():
        ez = np.zeros(10, np.int32); cig = np.zeros(ql + tl + 4, np.uint32); k = C.c_int()
        m = E.emu_ksw_extd2(ql, q, tl, t, mat, qq, e, q2, e2, w, zdrop, end_bonus, flag, force, ez, cig, len(cig), C.byref(k))
        klass[k.value] += 1
        if m == -1 and force >= 0: continue
        if m < 0 or [int(x) for x in ez] != [o[kk] for kk in W.EZ_FIELDS] or not np.array_equal(cig[:max(m,0)], o["cigar"]):
            bad += 1; print("MISMATCH", v, n, ql, tl, w, zdrop, end_bonus, hex(flag), k.value, flush=True)
    n += 1; cells += ql * tl
    if n % 500 == 0: print(n, "jobs", "%.0f s" % (time.time() - t0), "bad", bad, flush=True)
print(n, "jobs replayed through", variants, "; classes", dict(sorted(klass.items())), "; bad", bad)
