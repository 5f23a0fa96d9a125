"""Build the native libraries in-tree.

    python -m fatezero_amd.build            # libfatezero_hip.so  (hipcc, gfx950)  -- the product
    python -m fatezero_amd.build --emu      # libfatezero_emu.so  (host clang++)   -- test infrastructure only

The emulation build compiles the *same* kernel sources against the fiber-based CPU model in csrc/fz_rt.h /
csrc/fz_emu.cpp so that kernel index math and the host orchestration can be tested without a GPU.  The product
never loads it (fatezero_amd/_native.py only ever opens libfatezero_hip.so).
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIP_LIB = os.path.join(HERE, "libfatezero_hip.so")
EMU_LIB = os.path.join(HERE, "libfatezero_emu.so")


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def build_hip(force=False, verbose=False):
    srcs = _sources()
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    if not force and not _stale(HIP_LIB, srcs + hdrs):
        return HIP_LIB
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    objdir = os.path.join(HERE, "build", "hip")
    os.makedirs(objdir, exist_ok=True)

    def cc(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if force or _stale(obj, [src] + hdrs):
            _run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only",
                  "-c", src, "-o", obj])
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, srcs))
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB] + objs)
    if verbose:
        print("built", HIP_LIB)
    return HIP_LIB


def build_emu(force=False, verbose=False):
    srcs = _sources() + [os.path.join(CSRC, "fz_emu.cpp")]
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    if not force and not _stale(EMU_LIB, srcs + hdrs):
        return EMU_LIB
    cxx = os.path.join(ROCM, "lib", "llvm", "bin", "clang++")
    objdir = os.path.join(HERE, "build", "emu")
    os.makedirs(objdir, exist_ok=True)

    def cc(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if force or _stale(obj, [src] + hdrs):
            _run([cxx, "-x", "c++", "-DFZ_EMU", "-O2", "-std=c++17", "-fPIC", "-march=native", "-Wno-unknown-attributes",
                  "-Wno-unused-value", "-c", src, "-o", obj])
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, srcs))
    _run([cxx, "-shared", "-fPIC", "-o", EMU_LIB] + objs + ["-lpthread"])
    if verbose:
        print("built", EMU_LIB)
    return EMU_LIB


if __name__ == "__main__":
    force = "--force" in sys.argv
    if "--emu" in sys.argv:
        build_emu(force, True)
    else:
        build_hip(force, True)
