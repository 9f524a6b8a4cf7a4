"""Kernel logic on the CPU: the product kernel headers compiled against the host wavefront emulator
(tests/simt_emu) must agree bit-for-bit with the oracle. This is how kernel bugs are found without a GPU."""
import ctypes as C
import numpy as np
import pytest
import wmtest as W
import kswcases
from winnowmap_amd import build


@pytest.fixture(scope="module")
def emu():
    E = C.CDLL(build.build_emu())
    E.emu_ksw_extd2.argtypes = [C.c_int, W.u8p, C.c_int, W.u8p, W.i8p] + [C.c_int] * 9 + [W.i32p, W.u32p, C.c_int, C.POINTER(C.c_int)]
    return E


def emu_ksw(E, c, force=-1):
    ez = np.zeros(10, np.int32)
    cig = np.zeros(len(c["q"]) + len(c["t"]) + 4, np.uint32)
    k = C.c_int()
    n = E.emu_ksw_extd2(len(c["q"]), c["q"], len(c["t"]), c["t"], W.simple_mat(c["a"], c["b"], 1), c["q_"], c["e"], c["q2"], c["e2"],
                        c["w"], c["zdrop"], c["end_bonus"], c["flag"], force, ez, cig, len(cig), C.byref(k))
    return n, ez, cig[:max(n, 0)], k.value


@pytest.mark.parametrize("seed", [1, 2])
def test_ksw_emulated_kernel_matches_oracle(emu, seed):
    seen = set()
    for c in kswcases.make_cases(seed, 90, max_len=600):
        o = W.o_ksw_extd2(c["q"], c["t"], mat=W.simple_mat(c["a"], c["b"], 1), q=c["q_"], e=c["e"], q2=c["q2"], e2=c["e2"],
                          w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])
        for force in (-1, 3, 7, 11):          # host's choice, then the CLIP+HASN variant of every window size
            n, ez, cig, klass = emu_ksw(emu, c, force)
            if n < 0:
                continue
            seen.add(klass)
            assert [int(x) for x in ez] == [o[k] for k in W.EZ_FIELDS], (klass, c["flag"], c["w"])
            assert np.array_equal(cig, o["cigar"])
    assert len(seen) >= 6
