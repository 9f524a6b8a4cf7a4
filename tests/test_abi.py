"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/wm_gpu.h declares, and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re
import pytest
from winnowmap_amd import build, gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_gpu()
    return gpu.lib()


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "wm_gpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wm_[a-z0-9_]+)\s*\(", txt)))


def test_exports_every_declared_symbol(lib):
    syms = declared_symbols()
    assert len(syms) >= 10
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = lib.wm_ctx_create(0, 0, C.byref(h))
    assert rc == -1 and not h.value            # WM_ENODEV
    assert b"no HIP device" in lib.wm_last_error()
    with pytest.raises(gpu.WmError):
        gpu.Context(0)


def test_product_does_not_reference_oracle():
    # the product tree must never include / link / import anything under oracle/ or tests/
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "winnowmap_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                if f == "build.py":      # builds the checkers for the tests, never loads them
                    continue
                for line in open(os.path.join(d, f), errors="ignore"):
                    ls = line.strip()
                    if not (ls.startswith(("#include", "import ", "from ")) or "CDLL(" in ls or "dlopen(" in ls):
                        continue
                    for needle in ("oracle", "libwinnowmap_ref", "simt_emu", "wmtest"):
                        if needle in ls:
                            bad.append((f, ls))
    assert not bad, bad
