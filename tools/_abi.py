"""One calling convention over libso3proj.so builds of different rounds, for the A/B tools: round 3 exported plain / _ws / _acc
spellings of the reducing entry points, round 4 one *_v2 entry each (workspace nullable, a flags word)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poseestimation_amd import _lib

P, I64, U32, INT = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint32, ctypes.c_int


class Lib:
    def __init__(self, path):
        self.name = os.path.basename(path).replace("libso3proj_", "").replace(".so", "")
        lib = ctypes.CDLL(path)
        self.v2 = hasattr(lib, "so3_frob_fwd_bwd_v2_f32")
        def refused(rc, func, args):                # a timing of a call the library REFUSED is the host's error path, not a kernel's
            if rc != 0:
                why = ctypes.cast(lib.so3_last_error(), ctypes.c_char_p).value if hasattr(lib, "so3_last_error") else b"?"
                raise RuntimeError("%s returned %d (%s)" % (func.__name__, rc, why))
            return rc
        if hasattr(lib, "so3_last_error"):
            lib.so3_last_error.restype = ctypes.c_void_p
        sig = lambda name, args: (setattr(getattr(lib, name), "restype", INT), setattr(getattr(lib, name), "argtypes", args),
                                  setattr(getattr(lib, name), "errcheck", refused))
        sig("so3_project_fwd_f32", [P, P, P, I64, P])
        sig("so3_project_fwd_bf16", [P, P, P, I64, P])
        sig("so3_project_bwd_f32", [P, P, P, I64, P])
        if self.v2:
            sig("so3_frob_fwd_bwd_v2_f32", [P, P, P, P, P, P, P, U32, I64, P])
            sig("so3_frob_loss_v2_f32", [P, P, P, P, P, P, U32, I64, P])
            sig("so3_angle_error_v2", [P, P, P, P, P, P, U32, I64, P])
            sig("so3_project_angle_error_v2_f32", [P, P, P, P, P, P, P, U32, I64, P])
        else:
            sig("so3_frob_fwd_bwd_ws_f32", [P, P, P, P, P, P, P, I64, P])
            sig("so3_frob_fwd_bwd_f32", [P, P, P, P, P, I64, P])
            sig("so3_frob_loss_ws_f32", [P, P, P, P, P, P, I64, P])
            sig("so3_angle_error_acc", [P, P, P, P, P, INT, I64, P])
            sig("so3_project_angle_error_acc_f32", [P, P, P, P, P, P, INT, I64, P])
        self.lib = lib

    def k1(self, m, r, n, st, flip=None):
        return self.lib.so3_project_fwd_f32(m, r, flip, n, st)

    def k1_bf16(self, m, r, n, st):
        return self.lib.so3_project_fwd_bf16(m, r, None, n, st)

    def k2(self, m, g, dm, n, st):
        return self.lib.so3_project_bwd_f32(m, g, dm, n, st)

    def k3(self, m, t, r, dm, ls, lm, ws, n, st):
        """fused head + loss + backward; ws None: memset + atomics"""
        if self.v2:
            return self.lib.so3_frob_fwd_bwd_v2_f32(m, t, r, dm, ls, lm, ws, 0, n, st)
        if ws is None:
            return self.lib.so3_frob_fwd_bwd_f32(m, t, r, dm, ls, n, st)
        return self.lib.so3_frob_fwd_bwd_ws_f32(m, t, r, dm, ls, lm, ws, n, st)

    def k3p(self, p_, t, g, ls, lm, ws, n, st):
        if self.v2:
            return self.lib.so3_frob_loss_v2_f32(p_, t, g, ls, lm, ws, 0, n, st)
        return self.lib.so3_frob_loss_ws_f32(p_, t, g, ls, lm, ws, n, st)

    def k4_sum(self, a, b, sc, fl, n, st):
        """(sum, count) into pre-zeroed slots"""
        if self.v2:
            return self.lib.so3_angle_error_v2(a, b, None, sc, fl, None, _lib.PREZEROED, n, st)
        return self.lib.so3_angle_error_acc(a, b, None, sc, fl, 0, n, st)

    def k14_sum(self, m, t, sc, fl, n, st, exact=False):
        """fused head + metric, (sum, count) into pre-zeroed slots; round 3 is float64 on every row whatever `exact` says"""
        if self.v2:
            return self.lib.so3_project_angle_error_v2_f32(m, t, None, None, sc, fl, None, _lib.PREZEROED | (_lib.EXACT_F64 if exact else 0), n, st)
        return self.lib.so3_project_angle_error_acc_f32(m, t, None, None, sc, fl, 0, n, st)
