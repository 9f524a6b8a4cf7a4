#!/usr/bin/env python
"""VGPR / SGPR / spill / LDS per kernel from the device assembly of wm_gpu.hip (hipcc -S --cuda-device-only).
  python tools/kernel_regs.py [out.txt]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
asm = os.path.join(tmp, "wm.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "-w",
                       "--cuda-device-only", "-S", "-o", asm, os.path.join(ROOT, "winnowmap_amd", "csrc", "wm_gpu.hip")])
txt = open(asm).read()
meta = txt[txt.index("amdhsa.kernels:"):]
rows = []
for blk in re.split(r"\n  - \.agpr_count", meta)[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    v = int(g("vgpr_count"))
    alloc = (v + 7) // 8 * 8
    rows.append((name, v, min(8, 512 // max(alloc, 1)), g("sgpr_count"), g("vgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
out = "%-52s %5s %10s %5s %6s %8s %8s\n" % ("kernel", "vgpr", "waves/SIMD", "sgpr", "spill", "lds", "scratch")
for r in sorted(rows):
    out += "%-52s %5d %10d %5s %6s %8s %8s\n" % r
print(out)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(out)
