"""ctypes loader of oracle/kernel_model.cpp: the kernels' own templates compiled for the host.  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import shutil
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "kernel_model.cpp")
HDRS = [os.path.join(_HERE, "..", "poseestimation_amd", "csrc", h) for h in ("so3_device.h", "so3_rows.h")]
# SO3_SANITIZE=1: the same sources with -fsanitize=address,undefined -fno-sanitize-recover, into a library of its own (tools/sanitize_cpu.py
# drives it in a process that has the sanitizer's shared runtime preloaded; tests/test_sanitizers.py)
SANITIZE = os.environ.get("SO3_SANITIZE") == "1"
SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-shared-libasan", "-g", "-fno-omit-frame-pointer"]
LIB = os.path.join(_HERE, "libso3model_san.so" if SANITIZE else "libso3model.so")
_lib = None


def clangxx():
    for cand in (os.environ.get("CLANGXX"), "/opt/rocm/lib/llvm/bin/clang++", shutil.which("amdclang++"), shutil.which("clang++")):
        if cand and os.path.exists(cand):
            return cand
    return None


def build(force: bool = False) -> str:
    cxx = clangxx()
    if cxx is None:
        raise RuntimeError("kernel_model needs clang++ (ext_vector_type); none found")
    stale = force or not os.path.exists(LIB) or any(os.path.getmtime(f) > os.path.getmtime(LIB) for f in [SRC] + HDRS)
    if stale:
        extra = os.environ.get("SO3_MODEL_DEFINES", "").split()          # e.g. "-DSO3_QUAT_STALL=0": the A/B of a threshold, counted on the host
        if SANITIZE:
            extra += SAN_FLAGS
        subprocess.check_call([cxx, "-x", "c++", "-std=c++17", "-O1" if SANITIZE else "-O2", "-ffp-contract=off", "-fPIC", "-shared", *extra, "-o", LIB + ".tmp", SRC])
        os.replace(LIB + ".tmp", LIB)
    return LIB


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def _c(a, dt):
    return np.ascontiguousarray(np.asarray(a, dt).reshape(-1, 9))


def project_jacobi(m):
    """The Jacobi path alone (hard rows of K1, and what K2/K3 recompute)."""
    m = _c(m, np.float32)
    r = np.empty_like(m)
    lib().model_project_jacobi_f32(_p(m), _p(r), ctypes.c_int64(m.shape[0]))
    return r.reshape(-1, 3, 3)


def project_quat(m):
    """The quaternion fast path alone: (R, hard).  Rows with hard set are redone by the Jacobi path in the product."""
    m = _c(m, np.float32)
    r = np.empty_like(m)
    hard = np.zeros(m.shape[0], np.uint8)
    lib().model_project_quat_f32(_p(m), _p(r), _p(hard), ctypes.c_int64(m.shape[0]))
    return r.reshape(-1, 3, 3), hard.astype(bool)


def project(m, packed=False, want_flip=False):
    m = _c(m, np.float32)
    r = np.empty_like(m)
    flip = np.zeros(m.shape[0], np.uint8)
    if packed:
        lib().model_project_packed_f32(_p(m), _p(r), ctypes.c_int64(m.shape[0]))
    else:
        lib().model_project_f32(_p(m), _p(r), _p(flip), ctypes.c_int64(m.shape[0]))
    r = r.reshape(-1, 3, 3)
    return (r, flip.astype(bool)) if want_flip else r


def project_bwd_jacobi(m, g):
    """The backward through the Jacobi frames alone (hard rows of K2 / K3)."""
    m, g = _c(m, np.float32), _c(g, np.float32)
    d = np.empty_like(m)
    lib().model_project_bwd_jacobi_f32(_p(m), _p(g), _p(d), ctypes.c_int64(m.shape[0]))
    return d.reshape(-1, 3, 3)


def project_bwd(m, g):
    m, g = _c(m, np.float32), _c(g, np.float32)
    d = np.empty_like(m)
    lib().model_project_bwd_f32(_p(m), _p(g), _p(d), ctypes.c_int64(m.shape[0]))
    return d.reshape(-1, 3, 3)


def project_f64(m):
    m = _c(m, np.float64)
    r = np.empty_like(m)
    lib().model_project_f64(_p(m), _p(r), ctypes.c_int64(m.shape[0]))
    return r.reshape(-1, 3, 3)


def project_bwd_f64(m, g):
    m, g = _c(m, np.float64), _c(g, np.float64)
    d = np.empty_like(m)
    lib().model_project_bwd_f64(_p(m), _p(g), _p(d), ctypes.c_int64(m.shape[0]))
    return d.reshape(-1, 3, 3)


HEAD_WIDTH = {"quat": 4, "euler": 3, "ortho5d": 5, "expmap": 3, "ortho6d": 6}


def head(name, x):
    x = np.ascontiguousarray(np.asarray(x, np.float32).reshape(-1, HEAD_WIDTH[name]))
    r = np.empty((x.shape[0], 9), np.float32)
    getattr(lib(), "model_%s_fwd" % name)(_p(x), _p(r), ctypes.c_int64(x.shape[0]))
    return r.reshape(-1, 3, 3)


def head_bwd(name, x, g):
    x = np.ascontiguousarray(np.asarray(x, np.float32).reshape(-1, HEAD_WIDTH[name]))
    g = _c(g, np.float32)
    d = np.empty_like(x)
    getattr(lib(), "model_%s_bwd" % name)(_p(x), _p(g), _p(d), ctypes.c_int64(x.shape[0]))
    return d


def se3_update(out12, t_init, fx, fy):
    o = np.ascontiguousarray(np.asarray(out12, np.float32).reshape(-1, 12))
    t = np.ascontiguousarray(np.asarray(t_init, np.float32).reshape(-1, 16))
    r = np.empty_like(t)
    lib().model_se3_update(_p(o), _p(t), _p(r), ctypes.c_float(fx), ctypes.c_float(fy), ctypes.c_int64(o.shape[0]))
    return r.reshape(-1, 4, 4)


def se3_update_bwd(out12, t_init, g, fx, fy):
    o = np.ascontiguousarray(np.asarray(out12, np.float32).reshape(-1, 12))
    t = np.ascontiguousarray(np.asarray(t_init, np.float32).reshape(-1, 16))
    g = np.ascontiguousarray(np.asarray(g, np.float32).reshape(-1, 16))
    d = np.empty_like(o)
    lib().model_se3_update_bwd(_p(o), _p(t), _p(g), _p(d), ctypes.c_float(fx), ctypes.c_float(fy), ctypes.c_int64(o.shape[0]))
    return d


def park_reserve_interleaved(start, cap, n, nested=False):
    """so3_rows.h's reservation protocol under a replayed interleaving (see kernel_model.cpp): (final count, base per actor)."""
    n = np.ascontiguousarray(n, np.uint32)
    base = np.full(n.shape[0], -2, np.int32)
    f = lib().model_park_reserve_interleaved
    f.restype = ctypes.c_uint
    count = f(ctypes.c_uint(start), ctypes.c_uint(cap), _p(n), ctypes.c_int(n.shape[0]), ctypes.c_int(1 if nested else 0), _p(base))
    return int(count), base


def fast_path_counters(reset=True):
    """(rows whose first eigenvector was not final, adjugates computed for them) since the last reset."""
    out = np.zeros(2, np.int64)
    lib().model_fast_path_counters(_p(out), ctypes.c_int(1 if reset else 0))
    return int(out[0]), int(out[1])
