"""Build libso3proj.so (hipcc, gfx950) in-tree.  `python -m poseestimation_amd.build [--force]`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libso3proj.so")
SOURCES = ["so3proj.hip"]
HEADERS = ["so3_device.h", "so3_rows.h", os.path.join("..", "..", "include", "so3proj.h")]
# -fno-slp-vectorize: hipcc's SLP pass packs the 3-vector arithmetic into v_pk_fma_f32/v_pk_mul_f32,
# which on gfx950 issue at half the rate of the scalar forms (tools/ubench/valu_rates.hip: same
# FLOP/s) and cost ~250 extra v_mov_b32 per lane to build the register pairs.
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
               "-fno-slp-vectorize"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm); libso3proj.so cannot be built")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_fastcall(force: bool = False, verbose: bool = False):
    """The CPython entry for enqueue-only calls (csrc/fastcall.c), next to the package.  Optional: returns None when no C
    compiler or Python.h is available -- the mirror then calls through ctypes."""
    import sysconfig
    src = os.path.join(CSRC, "fastcall.c")
    out = os.path.join(_HERE, "_so3fast" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    cc = shutil.which("gcc") or shutil.which("cc")
    inc = sysconfig.get_paths().get("include")
    if cc is None or inc is None or not os.path.exists(os.path.join(inc, "Python.h")):
        return None
    cmd = [cc, "-O2", "-fPIC", "-shared", "-I", inc, "-o", out + ".tmp", src]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    try:
        subprocess.check_call(cmd)
    except (OSError, subprocess.CalledProcessError):
        return None
    os.replace(out + ".tmp", out)
    return out


def build_autograd_node(force: bool = False, verbose: bool = False):
    """frobenius_head's autograd node in C++ (csrc/autograd_node.cpp), next to the package: plain g++ against torch's headers and
    libraries (no device code, no HIP headers), ~40 s.  Optional: returns None when it cannot be built -- the mirror then uses
    its Python autograd.Function."""
    import sysconfig
    src = os.path.join(CSRC, "autograd_node.cpp")
    out = os.path.join(_HERE, "_so3node" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        return None
    try:
        import torch
        from torch.utils import cpp_extension
    except ImportError:
        return None
    incs = [i for i in cpp_extension.include_paths() if os.path.isdir(i)] + [sysconfig.get_paths()["include"]]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    rocm_inc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "include")     # c10/hip/HIPStream.h needs hip_runtime_api.h (host API only)
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-DTORCH_EXTENSION_NAME=_so3node", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM", "-Wno-deprecated-declarations",
           *["-isystem" + i for i in incs], "-isystem" + rocm_inc, "-o", out + ".tmp", src,
           "-L" + libdir, "-Wl,-rpath," + libdir, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch", "-ltorch_python"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    try:
        subprocess.check_call(cmd)
    except (OSError, subprocess.CalledProcessError):
        return None
    os.replace(out + ".tmp", out)
    return out


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 (cross-compiles without a GPU).  Returns the .so path."""
    build_fastcall(force, verbose)
    build_autograd_node(force, verbose)
    if not force and not is_stale():
        return LIB
    cmd = [hipcc(), *HIPCC_FLAGS, "-o", LIB + ".tmp", *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


def build_test_variant(name: str, defines, force: bool = False, verbose: bool = False) -> str:
    """libso3proj_<name>.so next to the library: the same sources with -D switches, for tests that need a code path the shipped
    constants make rare (spins0: SO3_STAT_SPINS=0 -- every finishing workgroup of so3_angle_stats times out at once -- and four workgroups per CU, so
    that most of its launch-mates start after it has finished).  Test
    infrastructure: nothing in the package loads it."""
    out = os.path.join(_HERE, "libso3proj_%s.so" % name)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    cmd = [hipcc(), *HIPCC_FLAGS, *["-D" + d for d in defines], "-o", out + ".tmp", *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


TEST_VARIANTS = {"spins0": ["SO3_STAT_SPINS=0", "SO3_STAT_GRID_MULT=4"]}


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
