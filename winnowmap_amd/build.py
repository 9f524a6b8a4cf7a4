"""Build the in-tree native artefacts: libwmgpu.so (hipcc, gfx950) and the test-side checkers."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwmgpu.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


class _Lock:
    """one build of a given artefact at a time, across processes (pytest-xdist workers ask for the same libraries at once)"""

    def __init__(self, target):
        self.path = target + ".lock"

    def __enter__(self):
        import fcntl
        self.f = open(self.path, "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)

    def __exit__(self, *a):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()


def _run_to(target, cmd_for):
    """run cmd_for(tmp) and move tmp over `target` only when it succeeded: nobody ever loads a half-written library"""
    tmp = target + ".tmp%d" % os.getpid()
    try:
        subprocess.check_call(cmd_for(tmp))
        os.replace(tmp, target)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


# -disable-promote-alloca-to-vector: the ksw kernels keep their per-lane state in small register arrays indexed by fully unrolled loops.
# AMDGPUPromoteAllocaToVector runs BEFORE the unroller, turns each array into one <N x i32> value, and every uniform branch that assigns an
# element then copies the whole vector at its join: 1536 v_mov_b64 (37 % of the VALU instructions) and 31 extra VGPRs in
# ksw_dpp_kernel<16,false,false,true>. Without the pass SROA splits the arrays into scalars after unrolling; no kernel of the library gains
# scratch (tools/kernel_regs.py; two rocPRIM sort kernels of the k-mer counter take 64 B). profiles/r02z_codegen_flag.txt
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "-mllvm", "-disable-promote-alloca-to-vector"]


UNITS = ("wm_rt", "wm_ksw", "wm_index", "wm_window", "wm_mapper")      # the translation units of libwmgpu.so (csrc/wm_rt.h says what each holds)


def build_gpu(force=False, verbose=False, out=None):
    """libwmgpu.so (or `out`: a variant library for A/B runs, selected with WM_LIBWMGPU=<path>; built here so that no GPU minute is spent compiling).
    The five units are compiled side by side (objects under csrc/.obj/<library name>/, git-ignored) and linked into the one shared object."""
    global LIB
    if out is not None:
        saved, LIB = LIB, out
        try:
            return build_gpu(force, verbose)
        finally:
            LIB = saved
    srcs = [os.path.join(d, f) for d, _, fs in os.walk(CSRC) for f in fs if ".obj" not in d] + [os.path.join(ROOT, "include", "wm_gpu.h")]
    # WM_KERNEL_DEFINES="WM_KSW_ROR=0 ...": kernel variants under evaluation (A/B on a GPU box); the default build defines nothing. The defines a
    # library was built with are part of its staleness check (sidecar stamp) and are compiled into it (wm_build_defines(), recorded by bench.py)
    defines = " ".join(os.environ.get("WM_KERNEL_DEFINES", "").split())
    stamp = LIB + ".defines"
    with _Lock(LIB):
        have = open(stamp).read() if os.path.exists(stamp) else ""
        if not force and not _newer(LIB, srcs) and have == defines:
            return LIB
        defs = ["-D" + d for d in defines.split()] + ['-DWM_BUILD_DEFINES="%s"' % defines]
        objdir = os.path.join(CSRC, ".obj", os.path.basename(LIB))
        os.makedirs(objdir, exist_ok=True)
        cmds = [[HIPCC] + HIP_FLAGS + defs + ["-fPIC", "-c", os.path.join(CSRC, u + ".hip"), "-o", os.path.join(objdir, u + ".o")] for u in UNITS]
        if verbose:
            for c in cmds:
                print(" ".join(c), file=sys.stderr)
        procs = [subprocess.Popen(c) for c in cmds]
        rcs = [p.wait() for p in procs]
        if any(rcs):
            raise subprocess.CalledProcessError(next(r for r in rcs if r), cmds[next(i for i, r in enumerate(rcs) if r)])
        link = lambda out_: [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out_] + [os.path.join(objdir, u + ".o") for u in UNITS] + ["-lz", "-lpthread"]  # noqa: E731
        if verbose:
            print(" ".join(link(LIB)), file=sys.stderr)
        _run_to(LIB, link)
        with open(stamp, "w") as f:
            f.write(defines)
    return LIB


def build_oracle():
    """oracle/libwm_oracle.so always; oracle/_ref only where /root/reference exists (the build container)."""
    with _Lock(os.path.join(ROOT, "oracle", "libwm_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], stdout=subprocess.DEVNULL)
        if os.path.exists("/root/reference/src/map.c"):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref", "-j8"], stdout=subprocess.DEVNULL)
            if os.path.exists(LIB):      # the reference's CLI bound to / substituted by libwmgpu.so (oracle/wm_binding.cpp, oracle/wm_subst.cpp)
                subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "wm", "subst", "-j8"], stdout=subprocess.DEVNULL)


def build_emu(defines=()):
    """tests/simt_emu/libwm_emu[_<defines>].so: the kernel headers compiled for the host against the wavefront emulator. `defines` selects
    kernel variants that are not the library default yet (e.g. ("WM_KSW_ROR=1",)), so that the tests cover them as well."""
    emu = os.path.join(ROOT, "tests", "simt_emu")
    tag = "".join("_" + "".join(ch if ch.isalnum() else "_" for ch in d) for d in defines)
    out = os.path.join(emu, "libwm_emu%s.so" % tag)
    srcs = [os.path.join(emu, f) for f in ("emu_driver.cpp", "simt.h")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    with _Lock(out):
        if _newer(out, srcs):
            _run_to(out, lambda o: ["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas"] +
                    ["-D" + d for d in defines] + ["-I" + emu, "-I" + CSRC, "-o", o, os.path.join(emu, "emu_driver.cpp")])
    return out


def build_emu_stripe(defines=()):
    """tests/simt_emu/libwm_emu_stripe[_<defines>].so: the stripe-pipelined ksw kernel alone on the emulator, with its event counters and the
    polling watchdog (tests/simt_emu/emu_stripe.cpp). ("WM_STRIPE_TEST_SLACK=...",) builds the variant whose bookkeeping margin is useless, so that
    the repeat-in-safe-mode path runs."""
    emu = os.path.join(ROOT, "tests", "simt_emu")
    tag = "".join("_" + "".join(ch if ch.isalnum() else "_" for ch in d) for d in defines)
    out = os.path.join(emu, "libwm_emu_stripe%s.so" % tag)
    srcs = [os.path.join(emu, f) for f in ("emu_stripe.cpp", "simt.h")] + \
           [os.path.join(CSRC, f) for f in ("ksw_stripe_kernel.h", "ksw_packed_kernel.h", "ksw_kernel.h", "ksw_plan.h")]
    with _Lock(out):
        if _newer(out, srcs):
            _run_to(out, lambda o: ["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-Wno-unknown-pragmas"] +
                    ["-D" + d for d in defines] + ["-I" + emu, "-I" + CSRC, "-o", o, os.path.join(emu, "emu_stripe.cpp")])
    return out


def build_emu_chain(defines=()):
    """tests/simt_emu/libwm_emu_chain[_<defines>].so: the chained-workgroup ksw kernel (ksw_chain_kernel.h) on the emulator, one host thread per wavefront,
    the mailbox in plain memory (tests/simt_emu/emu_chain.cpp)."""
    emu = os.path.join(ROOT, "tests", "simt_emu")
    tag = "".join("_" + "".join(ch if ch.isalnum() else "_" for ch in d) for d in defines)
    out = os.path.join(emu, "libwm_emu_chain%s.so" % tag)
    srcs = [os.path.join(emu, f) for f in ("emu_chain.cpp", "simt.h")] + \
           [os.path.join(CSRC, f) for f in ("ksw_chain_kernel.h", "ksw_stripe_kernel.h", "ksw_packed_kernel.h", "ksw_kernel.h", "ksw_plan.h")]
    with _Lock(out):
        if _newer(out, srcs):
            _run_to(out, lambda o: ["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-Wno-unknown-pragmas"] +
                    ["-D" + d for d in defines] + ["-I" + emu, "-I" + CSRC, "-o", o, os.path.join(emu, "emu_chain.cpp")])
    return out


def build_harness():
    """tests/host_harness/libwm_harness.so: the product HOST mapper driven by oracle-backed device ops (tests only)."""
    hd = os.path.join(ROOT, "tests", "host_harness")
    out = os.path.join(hd, "libwm_harness.so")
    srcs = [os.path.join(hd, "harness.cpp"), os.path.join(ROOT, "oracle", "wm_oracle.c")] + \
           [os.path.join(CSRC, "host", f) for f in os.listdir(os.path.join(CSRC, "host"))]
    with _Lock(out):
        if _newer(out, srcs):
            _run_to(out, lambda o: ["g++", "-std=c++17", "-O2", "-g", "-ffp-contract=off", "-fPIC", "-shared", "-o", o,
                                    os.path.join(hd, "harness.cpp"), os.path.join(ROOT, "oracle", "wm_oracle.c"), "-lz", "-pthread"])
    return out


if __name__ == "__main__":
    build_gpu(force="--force" in sys.argv, verbose=True)
    build_oracle()
    build_emu()
    build_emu_stripe()
    build_emu_chain()
    build_harness()
