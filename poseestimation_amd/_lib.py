"""ctypes binding of libso3proj.so (the C ABI declared in include/so3proj.h).

No CPU fallback: if the shared library is missing or a tensor is not on a HIP device the call
raises.  The library is looked up in-tree only (next to this file), so the driver sees the native
code that was actually loaded.
"""
from __future__ import annotations

import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libso3proj.so")

# name -> (restype, argtypes); must list every symbol include/so3proj.h declares.
_P, _I64, _I32, _INT, _U32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int, ctypes.c_uint32
RADIANS, PREZEROED, EXACT_F64 = 1, 2, 4          # include/so3proj.h: SO3_RADIANS, SO3_PREZEROED, SO3_EXACT_F64
SYMBOLS = {
    "so3_version": (_INT, []),
    "so3_last_error": (ctypes.c_char_p, []),
    "so3_last_kernel": (ctypes.c_char_p, []),
    "so3_project_fwd_f32": (_INT, [_P, _P, _P, _I64, _P]),
    "so3_project_fwd_bf16": (_INT, [_P, _P, _P, _I64, _P]),
    "so3_project_bwd_f32": (_INT, [_P, _P, _P, _I64, _P]),
    "so3_project_bwd_bf16": (_INT, [_P, _P, _P, _I64, _P]),
    "so3_project_fwd_f64": (_INT, [_P, _P, _P, _I64, _P]),
    "so3_project_bwd_f64": (_INT, [_P, _P, _P, _I64, _P]),
    "so3_frob_fwd_bwd_v2_f32": (_INT, [_P, _P, _P, _P, _P, _P, _P, _U32, _I64, _P]),
    "so3_frob_fwd_bwd_v2_bf16": (_INT, [_P, _P, _P, _P, _P, _P, _P, _U32, _I64, _P]),
    "so3_frob_loss_v2_f32": (_INT, [_P, _P, _P, _P, _P, _P, _U32, _I64, _P]),
    "so3_angle_error_v2": (_INT, [_P, _P, _P, _P, _P, _P, _U32, _I64, _P]),
    "so3_project_angle_error_v2_f32": (_INT, [_P, _P, _P, _P, _P, _P, _P, _U32, _I64, _P]),
    "so3_geodesic_f32": (_INT, [_P, _P, _P, _I64, _P]),
    "so3_geodesic_eps_f32": (_INT, [_P, _P, _P, _P, _P, _INT, ctypes.c_float, _P, _I64, _P]),
    "so3_angle_error_f64": (_INT, [_P, _P, _P, _P, _P, _INT, _P, _I64, _P]),
    "so3_geodesic_f64": (_INT, [_P, _P, _P, _I64, _P]),
    "so3_frob_loss_f64": (_INT, [_P, _P, _P, _P, _P, _P, _I64, _P]),
    "so3_project_fwd_diag_f32": (_INT, [_P, _P, _P, _I64, _P]),
    "so3_scale_f32": (_INT, [_P, _P, _P, _I64, _P]),
    "so3_scale_bf16": (_INT, [_P, _P, _P, _I64, _P]),
    "so3_reduce_workspace_bytes": (ctypes.c_size_t, []),
    "so3_se3_update_f32": (_INT, [_P, _P, _P, ctypes.c_float, ctypes.c_float, _I64, _P]),
    "so3_se3_update_bwd_f32": (_INT, [_P, _P, _P, _P, ctypes.c_float, ctypes.c_float, _I64, _P]),
    "so3_ortho6d_fwd_f32": (_INT, [_P, _P, _I64, _P]),
    "so3_ortho6d_bwd_f32": (_INT, [_P, _P, _P, _I64, _P]),
    **{"so3_%s_fwd_f32" % h: (_INT, [_P, _P, _I64, _P]) for h in ("quat", "euler", "ortho5d", "expmap")},
    **{"so3_%s_bwd_f32" % h: (_INT, [_P, _P, _P, _I64, _P]) for h in ("quat", "euler", "ortho5d", "expmap")},
    "so3_rotate_clouds_f32": (_INT, [_P, _P, _P, _INT, _I64, _I32, _P]),
    "so3_pc_normalize_f32": (_INT, [_P, _P, _P, _P, _I64, _I32, _P]),
    "so3_add_l1_f32": (_INT, [_P, _P, _P, _P, _P, _P, ctypes.c_float, _I64, _I32, _P]),
    "so3_add_l1_disentangled_f32": (_INT, [_P, _P, _P, _P, _P, ctypes.c_float, _I64, _I32, _P]),
    "so3_angle_stats_workspace_bytes": (ctypes.c_size_t, []),
    "so3_angle_stats": (_INT, [_P, _P, _I32, _P, _P, _I64, _P]),
    "so3_kabsch_f32": (_INT, [_P, _P, _P, _P, _I64, _I32, _P]),
    "so3_rotations_axis_angle_f32": (_INT, [_P, _P, _P, _I64, _P]),
    "so3_kabsch_synth_f32": (_INT, [_P, _P, ctypes.c_float, ctypes.c_uint32, _P, _P, _I64, _I32, _P]),
}

def _install_legacy(lib) -> None:
    """The round-3 spellings of the reducing entry points -- `static inline` wrappers in include/so3proj.h, not exports -- as
    attributes of the loaded library, argument for argument like the header's, so that scripts written against them keep
    running for this round.  The mirror itself calls the *_v2 functions."""
    rad = lambda r: RADIANS if r else 0
    f32, bf16, loss, ang, pang = (lib.so3_frob_fwd_bwd_v2_f32, lib.so3_frob_fwd_bwd_v2_bf16, lib.so3_frob_loss_v2_f32, lib.so3_angle_error_v2,
                                  lib.so3_project_angle_error_v2_f32)
    legacy = {
        "so3_frob_fwd_bwd_f32": lambda m, t, r, dm, ls, b, st: f32(m, t, r, dm, ls, None, None, 0, b, st),
        "so3_frob_fwd_bwd_bf16": lambda m, t, r, dm, ls, b, st: bf16(m, t, r, dm, ls, None, None, 0, b, st),
        "so3_frob_fwd_bwd_ws_f32": lambda m, t, r, dm, ls, lm, ws, b, st: f32(m, t, r, dm, ls, lm, ws, 0, b, st),
        "so3_frob_fwd_bwd_ws_bf16": lambda m, t, r, dm, ls, lm, ws, b, st: bf16(m, t, r, dm, ls, lm, ws, 0, b, st),
        "so3_frob_loss_f32": lambda p, t, g, ls, b, st: loss(p, t, g, ls, None, None, 0, b, st),
        "so3_frob_loss_ws_f32": lambda p, t, g, ls, lm, ws, b, st: loss(p, t, g, ls, lm, ws, 0, b, st),
        "so3_angle_error": lambda a, b_, d, sc, fl, r, b, st: ang(a, b_, d, sc, fl, None, rad(r), b, st),
        "so3_angle_error_ws": lambda a, b_, d, sc, fl, r, ws, b, st: ang(a, b_, d, sc, fl, ws, rad(r), b, st),
        "so3_angle_error_acc": lambda a, b_, d, sc, fl, r, b, st: ang(a, b_, d, sc, fl, None, rad(r) | PREZEROED, b, st),
        "so3_project_angle_error_f32": lambda m, t, r_, d, sc, fl, r, b, st: pang(m, t, r_, d, sc, fl, None, rad(r) | EXACT_F64, b, st),
        "so3_project_angle_error_ws_f32": lambda m, t, r_, d, sc, fl, r, ws, b, st: pang(m, t, r_, d, sc, fl, ws, rad(r) | EXACT_F64, b, st),
        "so3_project_angle_error_acc_f32": lambda m, t, r_, d, sc, fl, r, b, st: pang(m, t, r_, d, sc, fl, None, rad(r) | PREZEROED | EXACT_F64, b, st),
    }
    for name, fn in legacy.items():
        setattr(lib, name, fn)


LEGACY_INLINE = ("so3_frob_fwd_bwd_f32", "so3_frob_fwd_bwd_bf16", "so3_frob_fwd_bwd_ws_f32", "so3_frob_fwd_bwd_ws_bf16", "so3_frob_loss_f32",
                 "so3_frob_loss_ws_f32", "so3_angle_error", "so3_angle_error_ws", "so3_angle_error_acc", "so3_project_angle_error_f32",
                 "so3_project_angle_error_ws_f32", "so3_project_angle_error_acc_f32")

_lock = threading.Lock()
_lib = None


class So3ProjError(RuntimeError):
    """A C-ABI call returned non-zero."""


def load():
    """dlopen libso3proj.so and declare signatures.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f"{LIB_PATH} not found: the HIP extension has not been built. "
                    "Run `python -m poseestimation_amd.build` (needs hipcc; cross-compiles gfx950 "
                    "without a GPU). There is no CPU fallback.")
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SYMBOLS.items():
                fn = getattr(lib, name)      # AttributeError here = header/library mismatch
                fn.restype = res
                fn.argtypes = args
            _install_legacy(lib)
            _lib = lib
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().so3_last_error().decode("utf-8", "replace")
        raise So3ProjError(f"{what} failed with code {code}: {msg}")
