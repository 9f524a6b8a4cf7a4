#!/usr/bin/env python
"""VGPR / SGPR / spill / LDS / scratch and static instruction counts per kernel, from the device assembly of the library's units (wm_ksw.hip, wm_index.hip, wm_window.hip)
(hipcc -S --cuda-device-only with the library's flags, winnowmap_amd/build.py HIP_FLAGS).
  python tools/kernel_regs.py [out.txt]
  python tools/kernel_regs.py --compare [out.txt]     the same table without / with -mllvm -disable-promote-alloca-to-vector"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from winnowmap_amd.build import HIP_FLAGS  # noqa: E402

PROMOTE = ["-mllvm", "-disable-promote-alloca-to-vector"]


def device_asm(flags):
    """the device assembly of the library's kernel-bearing units (winnowmap_amd/build.py UNITS), one after the other"""
    tmp, txt = tempfile.mkdtemp(), ""
    for u in ("wm_ksw", "wm_index", "wm_window"):
        asm = os.path.join(tmp, u + ".s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-w", "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", asm,
                               os.path.join(ROOT, "winnowmap_amd", "csrc", u + ".hip")])
        txt += open(asm).read() + "\n"
    return txt


def parse(txt):
    """{mangled name: dict(vgpr, sgpr, spill, lds, scratch, valu, salu, mov64)}"""
    out = {}
    for blk in [b for meta in txt.split("amdhsa.kernels:")[1:] for b in re.split(r"\n  - \.agpr_count", meta.split("amdhsa.target:")[0])[1:]]:      # (one metadata section per unit)
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "0"])[1]
        out[g("name")] = dict(vgpr=int(g("vgpr_count")), sgpr=int(g("sgpr_count")), spill=int(g("vgpr_spill_count")),
                              lds=int(g("group_segment_fixed_size")), scratch=int(g("private_segment_fixed_size")), valu=0, salu=0, mov64=0)
    cur = None
    for line in txt.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = out.get(m.group(1))
        elif "s_endpgm" in line:
            cur = None
        elif cur is not None:
            if line.startswith("\tv_"):
                cur["valu"] += 1
                cur["mov64"] += "v_mov_b64" in line
            elif line.startswith("\ts_"):
                cur["salu"] += 1
    return out


def pretty(names):
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return {n: re.sub(r"\(.*", "", d).replace("void ", "") for n, d in zip(names, dem)}


def waves(v):
    return min(8, 512 // max((v + 7) // 8 * 8, 1))


args = [a for a in sys.argv[1:] if not a.startswith("--")]
if "--compare" in sys.argv:
    base = [f for f in HIP_FLAGS if f not in PROMOTE]
    a, b = parse(device_asm(base)), parse(device_asm(base + PROMOTE))
    nm = pretty(list(a))
    out = "static per-kernel figures of wm_gpu.hip without -> with  -mllvm -disable-promote-alloca-to-vector  (VALU / v_mov_b64 = instructions in the code object)\n"
    out += "%-48s %11s %11s %13s %13s %11s\n" % ("kernel", "vgpr", "waves/SIMD", "VALU", "v_mov_b64", "scratch B")
    for k in sorted(a, key=lambda k: nm[k]):
        if "rocprim" in nm[k] and a[k]["scratch"] == b[k]["scratch"]:
            continue
        x, y = a[k], b[k]
        out += "%-48s %4d ->%4d %4d ->%4d %5d ->%5d %5d ->%5d %4d ->%4d\n" % (nm[k][:48], x["vgpr"], y["vgpr"], waves(x["vgpr"]), waves(y["vgpr"]),
                                                                           x["valu"], y["valu"], x["mov64"], y["mov64"], x["scratch"], y["scratch"])
else:
    a = parse(device_asm(HIP_FLAGS))
    nm = pretty(list(a))
    out = "%-52s %5s %10s %5s %6s %8s %8s %6s %6s\n" % ("kernel", "vgpr", "waves/SIMD", "sgpr", "spill", "lds", "scratch", "VALU", "SALU")
    for k in sorted(a, key=lambda k: nm[k]):
        if "rocprim" in nm[k]:
            continue
        x = a[k]
        out += "%-52s %5d %10d %5d %6d %8d %8d %6d %6d\n" % (nm[k][:52], x["vgpr"], waves(x["vgpr"]), x["sgpr"], x["spill"], x["lds"], x["scratch"], x["valu"], x["salu"])
print(out)
if args:
    open(args[0], "w").write(out)
