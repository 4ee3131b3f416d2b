"""ctypes front end of oracle/libso3oracle.so (the C float64 oracle).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SANITIZE = os.environ.get("SO3_SANITIZE") == "1"          # see oracle/kernel_model.py
_SO = os.path.join(_HERE, "libso3oracle_san.so" if SANITIZE else "libso3oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "so3_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        if SANITIZE:
            from . import kernel_model
            cc = kernel_model.clangxx().replace("clang++", "clang")          # ONE sanitizer runtime per process: clang's, as the model's
            subprocess.check_call([cc, "-std=c11", "-O1", "-fPIC", "-Wall", "-Wextra", "-ffp-contract=off", *kernel_model.SAN_FLAGS, "-shared", "-o", _SO, src, "-lm"])
        else:
            subprocess.check_call(["make", "-s", "-C", _HERE, "libso3oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        i64, i32, p = ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p
        _lib.oracle_project_f64.argtypes = [p, p, p, i64]
        _lib.oracle_project_f32.argtypes = [p, p, p, i64]
        _lib.oracle_project_bwd_f32.argtypes = [p, p, p, i64]
        _lib.oracle_angle_error.argtypes = [p, p, p, i64]
        _lib.oracle_angle_error.restype = ctypes.c_int
        _lib.oracle_kabsch_f32.argtypes = [p, p, p, p, i64, i32]
    return _lib


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def project(m, want_flip=False):
    """(B,3,3)/(B,9) float32 or float64 -> R of the same dtype (float64 arithmetic inside)."""
    m = np.asarray(m)
    dt = np.float64 if m.dtype == np.float64 else np.float32
    m = _c(m, dt).reshape(-1, 9)
    r = np.empty_like(m)
    flip = np.empty(m.shape[0], np.uint8)
    fn = lib().oracle_project_f64 if dt == np.float64 else lib().oracle_project_f32
    fn(m.ctypes.data, r.ctypes.data, flip.ctypes.data, m.shape[0])
    r = r.reshape(-1, 3, 3)
    return (r, flip.astype(bool)) if want_flip else r


def project_bwd(m, g):
    m = _c(m, np.float32).reshape(-1, 9)
    g = _c(g, np.float32).reshape(-1, 9)
    out = np.empty_like(m)
    lib().oracle_project_bwd_f32(m.ctypes.data, g.ctypes.data, out.ctypes.data, m.shape[0])
    return out.reshape(-1, 3, 3)


def angle_error(r1, r2):
    r1 = _c(r1, np.float32).reshape(-1, 9)
    r2 = _c(r2, np.float32).reshape(-1, 9)
    deg = np.empty(r1.shape[0], np.float64)
    bad = lib().oracle_angle_error(r1.ctypes.data, r2.ctypes.data, deg.ctypes.data, r1.shape[0])
    return deg, bool(bad)


def kabsch(p, q, want_h=False):
    p = _c(p, np.float32)
    q = _c(q, np.float32)
    b, n, _ = p.shape
    r = np.empty((b, 3, 3), np.float32)
    h = np.empty((b, 3, 3), np.float64)
    lib().oracle_kabsch_f32(p.ctypes.data, q.ctypes.data, r.ctypes.data, h.ctypes.data, b, n)
    return (r, h) if want_h else r
