"""Host-side mirror of the reference's hot-path interface, backed by libso3proj.so.

Same names, argument meaning and error behaviour as the reference functions they replace
(paths relative to the reference repository root):

    symmetric_orthogonalization(x)                     rotation_representation.py:192-206
    compute_geodesic_distance_from_two_matrices(m1,m2) rotation_representation.py:209-227
    angle_error(t_R1, t_R2)                            rotation_representation.py:230-242
    loss_frobenius(R_pred, R_true)                     3D-Pose/loss.py:7-11
    transform_output                                   rotation_representation.py:323-324

plus two fused entry points the reference spells as several calls:

    frobenius_head(x, R_true)      head + loss (+ backward in the same launch)   3D-Pose/main.py:60,85,90
    kabsch_rotation(P, Q)          bmm(Q^T, P) + head                            SURVEY.md section 8 a7

PyTorch is plumbing here: it owns device memory, the stream and autograd bookkeeping.  All
arithmetic happens in hand-written gfx950 kernels behind the C ABI (include/so3proj.h).  There is
no CPU path: a CPU tensor, or a missing libso3proj.so, raises.
"""
from __future__ import annotations

import torch
from torch.autograd.function import once_differentiable

from . import _lib

_RANGE_MSG = "angle out of range, input probably not proper rotation matrices"   # rotation_representation.py:238-239


# --------------------------------------------------------------------------------------------
# plumbing
# --------------------------------------------------------------------------------------------
# The reference's real batch sizes are 64-512 (Iterative/main.py:216, UPNA/main.py:126, 3D-Pose/configs/example.yaml:3): the
# kernels then take 3-4 us and everything in this section is on the critical path.  Hence: the library handle and its entry
# points are looked up once, pointers and the stream travel as plain ints (ctypes converts them through the argtypes declared
# in _lib.py), the raw stream comes from torch's own accessor, and the device guard is a no-op on a one-GPU process.
_L = None


def _libh():
    global _L
    if _L is None:
        _L = _lib.load()
    return _L


def _optional_helper(name: str, without: str):
    """An optional host-side helper module of this package, or None -- with ONE warning on stderr saying what is lost: both
    helpers only remove host overhead (results are the same bits either way), so a missing one must not fail the import, but it
    must not go unnoticed either (_so3node is compiled against the build machine's torch; another torch on the box where it
    runs silently moved config #4's step from the C++ nodes to the Python classes in round 3)."""
    import importlib
    try:
        return importlib.import_module("." + name, __package__)
    except ImportError as exc:
        import sys
        print("[poseestimation_amd] optional helper %s is not available (%s): %s.  "
              "`python -m poseestimation_amd.build --force` rebuilds it." % (name, exc, without), file=sys.stderr)
        return None


# csrc/fastcall.c: METH_FASTCALL entry for the enqueue-only calls (ctypes: ~2.5 us per call)
_so3fast = _optional_helper("_so3fast", "every C-ABI call goes through ctypes, ~2.5 us per call slower")
_FAST = {}


def _fn(name: str):
    """The C-ABI entry point `name` as a callable taking plain ints / None (pointers, sizes, the stream) and returning the int
    status: through _so3fast when it is built, else the ctypes function (argtypes declared in _lib.py).  Integer arguments only."""
    f = _FAST.get(name)
    if f is None:
        cfn = getattr(_libh(), name)
        if _so3fast is not None:
            import ctypes
            import functools
            f = functools.partial(_so3fast.call, ctypes.cast(cfn, ctypes.c_void_p).value)
        else:
            f = cfn
        _FAST[name] = f
    return f


# csrc/autograd_node.cpp: the autograd nodes without the interpreter in forward / backward
_so3node = _optional_helper("_so3node", "the Python autograd.Function classes serve every case, ~10-25 us per training step slower at batch 512")
_NODE_BOUND = False


def _node():
    """_so3node with the C-ABI addresses handed over (once), or None."""
    global _NODE_BOUND
    if _so3node is not None and not _NODE_BOUND:
        import ctypes
        lib = _libh()
        addr = lambda name: ctypes.cast(getattr(lib, name), ctypes.c_void_p).value
        _so3node.bind({name: addr(name) for name in (
            "so3_frob_fwd_bwd_v2_f32", "so3_frob_fwd_bwd_v2_bf16", "so3_scale_f32", "so3_scale_bf16", "so3_project_fwd_f32", "so3_project_fwd_bf16",
            "so3_project_bwd_f32", "so3_project_bwd_bf16", "so3_frob_loss_v2_f32", "so3_last_error")}, _SMALL_BATCH)
        _NODE_BOUND = True
    return _so3node


_ROW_HEADS = {}


def _row_head(symbol: str, width: int, x):
    """(..., width) -> (..., 3, 3) through the C++ node of the row-operation heads, or None when it (or its case) is not there."""
    node = _node()
    if node is None or type(x) is not torch.Tensor or not x.is_cuda:
        return None
    fns = _ROW_HEADS.get(symbol)
    if fns is None:
        import ctypes
        lib = _libh()
        fns = tuple(ctypes.cast(getattr(lib, "so3_%s_%s_f32" % (symbol, d)), ctypes.c_void_p).value for d in ("fwd", "bwd"))
        _ROW_HEADS[symbol] = fns
    return node.row_head(x, width, fns[0], fns[1], _stream(x.device))


def _require_device(*tensors: torch.Tensor) -> torch.device:
    dev = None
    for t in tensors:
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"expected a torch.Tensor, got {type(t).__name__}")
        if not t.is_cuda:
            raise RuntimeError(
                "poseestimation_amd runs on a HIP device only (tensor is on '%s'); there is no CPU "
                "fallback -- move the tensor to the MI355X with .cuda()" % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} and {t.device}")
    return dev


def _ptr(t):
    return t.data_ptr() if t is not None else None


class _NoGuard:
    __slots__ = ()

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()
_DEVICE_COUNT = None


def _on_device(dev: torch.device):
    """`with torch.cuda.device(dev)` only when dev is not already current (the context manager costs ~4 us per call)."""
    global _DEVICE_COUNT
    if _DEVICE_COUNT is None:
        _DEVICE_COUNT = torch.cuda.device_count()
    if _DEVICE_COUNT <= 1 or dev.index is None or dev.index == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(dev)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(dev: torch.device) -> int:
    """The current stream of `dev` as the integer the C ABI takes (hipStream_t)."""
    if _raw_stream is not None:
        return _raw_stream(dev.index if dev.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(dev).cuda_stream


_WORKSPACES = {}


def _workspace(dev: torch.device, stream: int):
    """The reduction workspace of (device, stream) -- include/so3proj.h: zero-filled once, then owned by that stream's calls.
    None while the stream is being captured into a graph (a replay may run beside eager calls: the no-workspace path then)."""
    if _capturing(dev):
        return None
    key = (dev.index, stream)
    ws = _WORKSPACES.get(key)
    if ws is None:
        ws = torch.zeros((_libh().so3_reduce_workspace_bytes(),), dtype=torch.uint8, device=dev)
        _WORKSPACES[key] = ws
    return ws


def _capturing(dev: torch.device) -> bool:
    """Is the current stream OF `dev` being captured into a graph?  (torch's query looks at the current device.)"""
    if dev.index is None or dev.index == torch.cuda.current_device():
        return torch.cuda.is_current_stream_capturing()
    with torch.cuda.device(dev):
        return torch.cuda.is_current_stream_capturing()


class _ZeroPool:
    """Zero-filled accumulator slots for the metric kernels (SO3_PREZEROED: sum_count[0] and the range flag must be 0 on entry):
    one torch.zeros per 256 calls instead of an init launch in front of every kernel.  A slot is handed out once; the tensors
    returned to the caller are views of it and keep their pool alive.
    One pool per (device, STREAM), like the reduction workspaces: the zero-fill is enqueued on the stream that was current when
    the pool was made, and a kernel on another stream could otherwise add into a slot before it has been zeroed -- or into a
    retired pool's block after the allocator handed it to someone else on the filling stream."""
    SLOTS = 256

    def __init__(self):
        self.pools = {}

    def take(self, dev: torch.device, stream: int):
        """(sum_count: 2 float64, flag: 1 int32), both zero, for a kernel enqueued on `stream` (the current stream of dev)."""
        if _capturing(dev):
            z = torch.zeros((1, 4), dtype=torch.float64, device=dev)               # a graph keeps its own (captured) zero-fill
            return z[0, :2], z[0, 2:3].view(torch.int32)[:1]
        key = (dev.index, stream)
        entry = self.pools.get(key)
        if entry is None or entry[1] >= self.SLOTS:
            with _on_device(dev):
                entry = [torch.zeros((self.SLOTS, 4), dtype=torch.float64, device=dev), 0]     # filled on `stream`: it is current
            self.pools[key] = entry
        row = entry[0][entry[1]]
        entry[1] += 1
        return row[:2], row[2:3].view(torch.int32)[:1]


_ZERO_POOL = _ZeroPool()
_SMALL_BATCH = 1024          # csrc: kSmallBatch -- up to here a reduction is one workgroup and needs no workspace


def _as_blocks(x: torch.Tensor) -> torch.Tensor:
    """x.view(-1, 3, 3) semantics (rotation_representation.py:199) -> contiguous (B, 9)."""
    if x.dim() == 2 and x.shape[1] == 9 and x.is_contiguous():
        return x
    if x.numel() % 9 != 0:
        raise RuntimeError(f"shape '[-1, 3, 3]' is invalid for input of size {x.numel()}")
    return x.reshape(-1, 9).contiguous()


def _head_input(x: torch.Tensor) -> torch.Tensor:
    """Kernel-ready (B,9) tensor: float32, bfloat16 kept as stored (math is fp32 in registers), or float64
    (its own float64 kernels, as the reference's function accepts double tensors)."""
    m = _as_blocks(x)
    dt = m.dtype
    if dt is torch.float32 or dt is torch.bfloat16 or dt is torch.float64:
        return m
    if dt is torch.float16:
        return m.float()
    raise TypeError(
        f"symmetric_orthogonalization: unsupported dtype {m.dtype} (inputs: float32, bfloat16, float16, float64)")


def _f32_blocks(t: torch.Tensor) -> torch.Tensor:
    m = _as_blocks(t)
    return m if m.dtype is torch.float32 else m.float()


_HEAD_FNS = {}


def _head_fns(dtype):
    """(forward, backward) entry points and the dtype the rotation / upstream gradient travel in."""
    fns = _HEAD_FNS.get(dtype)
    if fns is None:
        if dtype == torch.bfloat16:
            fns = (_fn("so3_project_fwd_bf16"), _fn("so3_project_bwd_bf16"), torch.float32)
        elif dtype == torch.float64:
            fns = (_fn("so3_project_fwd_f64"), _fn("so3_project_bwd_f64"), torch.float64)
        else:
            fns = (_fn("so3_project_fwd_f32"), _fn("so3_project_bwd_f32"), torch.float32)
        _HEAD_FNS[dtype] = fns
    return fns


def _no_double_backward(*grads) -> None:
    """torch's once_differentiable, at a fraction of its price (a no_grad context per call): these backward functions launch
    kernels autograd cannot see, so a backward whose result would have to be differentiated AGAIN must fail loudly.  That is
    once_differentiable's own condition -- grad mode on (create_graph=True) AND an incoming gradient that requires grad.
    create_graph=True alone (a gradient penalty on another branch of the graph) runs the kernels as always; nothing is recorded,
    the result is a constant."""
    if torch.is_grad_enabled() and any(g is not None and g.requires_grad for g in grads):
        raise RuntimeError("trying to differentiate twice a function that was marked with @once_differentiable "
                           "(poseestimation_amd kernels do not support double backward; the reference never uses it)")


def _check(code: int, what: str) -> None:
    if code != 0:
        # a call that failed may have enqueued part of its launches: the cached workspaces (include/so3proj.h: "re-zero it after a
        # call that returned an error") are dropped, the next call zero-fills fresh ones
        _WORKSPACES.clear()
        _STAT_WORKSPACES.clear()
        _lib.check(code, what)


# --------------------------------------------------------------------------------------------
# K1 / K2: the head
# --------------------------------------------------------------------------------------------
class _SymmetricOrthogonalization(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        dev = x.device if x.is_cuda else _require_device(x)
        m = _head_input(x)
        b = m.shape[0]
        fn, _, out_dtype = _head_fns(m.dtype)
        r = torch.empty((b, 3, 3), dtype=out_dtype, device=dev)
        with _on_device(dev):
            _check(fn(m.data_ptr(), r.data_ptr(), None, b, _stream(dev)), "so3_project_fwd")
        ctx.save_for_backward(m)
        ctx.in_shape = x.shape
        ctx.in_dtype = x.dtype
        return r

    @staticmethod
    def backward(ctx, grad_r):
        _no_double_backward(grad_r)
        (m,) = ctx.saved_tensors
        dev = m.device
        _, fn, g_dtype = _head_fns(m.dtype)
        g = grad_r.reshape(-1, 9)
        if not g.is_contiguous():
            g = g.contiguous()
        if g.dtype is not g_dtype:
            g = g.to(g_dtype)
        b = m.shape[0]
        dm = torch.empty_like(m)
        with _on_device(dev):
            _check(fn(m.data_ptr(), g.data_ptr(), dm.data_ptr(), b, _stream(dev)), "so3_project_bwd")
        if dm.dtype is not ctx.in_dtype:
            dm = dm.to(ctx.in_dtype)
        return dm.view(ctx.in_shape)


def symmetric_orthogonalization(x: torch.Tensor) -> torch.Tensor:
    """Maps 9D input vectors onto SO(3) via symmetric orthogonalization (SVD).

    x: [batch_size, 9] (any shape whose numel is a multiple of 9, as `x.view(-1, 3, 3)` accepts).
    Returns [batch_size, 3, 3] rotations R = U diag(1,1,det(UV^T)) V^T, differentiable: float32 for float32,
    bfloat16 and float16 input, float64 for float64 input.
    """
    if isinstance(x, torch.Tensor) and not (x.requires_grad and torch.is_grad_enabled()):
        # inference / evaluation loops: no autograd node to build
        dev = x.device if x.is_cuda else _require_device(x)
        m = _head_input(x)
        fn, _, out_dtype = _head_fns(m.dtype)
        b = m.shape[0]
        r = torch.empty((b, 3, 3), dtype=out_dtype, device=dev)
        with _on_device(dev):
            _check(fn(m.data_ptr(), r.data_ptr(), None, b, _stream(dev)), "so3_project_fwd")
        return r
    node = _node()
    if node is not None and type(x) is torch.Tensor and x.is_cuda:
        r = node.symmetric_orthogonalization(x, _stream(x.device))          # the C++ node; None for what it does not cover
        if r is not None:
            return r
    return _SymmetricOrthogonalization.apply(x)


def symmetric_orthogonalization_with_flip(x: torch.Tensor):
    """(R, flip): flip[b] is True where det(U V^T) < 0, the sign the reference multiplies into the
    last row of V^T (rotation_representation.py:202-204).  Not differentiable."""
    dev = _require_device(x)
    m = _head_input(x.detach())
    b = m.shape[0]
    fn, _, out_dtype = _head_fns(m.dtype)
    r = torch.empty((b, 3, 3), dtype=out_dtype, device=dev)
    flip = torch.empty((b,), dtype=torch.uint8, device=dev)
    with _on_device(dev):
        _check(fn(_ptr(m), _ptr(r), _ptr(flip), b, _stream(dev)), "so3_project_fwd")
    return r, flip.bool()


# --------------------------------------------------------------------------------------------
# K4: metrics
# --------------------------------------------------------------------------------------------
def _angle_call(r1, r2, want_deg, want_sum, radians=False):
    dev = _require_device(r1, r2)
    a, b_ = _f32_blocks(r1), _f32_blocks(r2)
    if a.shape != b_.shape:
        raise RuntimeError(f"angle_error: shape mismatch {tuple(r1.shape)} vs {tuple(r2.shape)}")
    n = a.shape[0]
    deg = torch.empty((n,), dtype=torch.float64, device=dev) if want_deg else None
    with _on_device(dev):
        st = _stream(dev)
        sc, flag = _ZERO_POOL.take(dev, st)         # zero-filled slots: the kernel needs no init launch in front of it
        _check(_fn("so3_angle_error_v2")(a.data_ptr(), b_.data_ptr(), _ptr(deg), sc.data_ptr() if want_sum else None, flag.data_ptr(), None,
                                          _lib.PREZEROED | (_lib.RADIANS if radians else 0), n, st), "so3_angle_error")
    return deg, (sc if want_sum else None), flag


def _is_f64(*ts) -> bool:
    return any(isinstance(t, torch.Tensor) and t.dtype == torch.float64 for t in ts)


def _f64_blocks(t: torch.Tensor) -> torch.Tensor:
    m = _as_blocks(t)
    return m if m.dtype is torch.float64 else m.double()


def _angle_call_f64(r1, r2, want_rows, want_sum, radians=False, geodesic=False):
    """float64 arguments (the reference casts to float64 before the product, rotation_representation.py:232-233, and returns
    the arguments' dtype from the geodesic distance): the same kernels' arithmetic on float64 data, so3_*_f64."""
    dev = _require_device(r1, r2)
    a, b_ = _f64_blocks(r1), _f64_blocks(r2)
    if a.shape != b_.shape:
        raise RuntimeError(f"angle_error: shape mismatch {tuple(r1.shape)} vs {tuple(r2.shape)}")
    n = a.shape[0]
    rows = torch.empty((n,), dtype=torch.float64, device=dev) if want_rows else None
    sc = torch.empty((2,), dtype=torch.float64, device=dev) if want_sum else None
    flag = None if geodesic else torch.empty((1,), dtype=torch.int32, device=dev)
    with _on_device(dev):
        if geodesic:
            _check(_libh().so3_geodesic_f64(a.data_ptr(), b_.data_ptr(), rows.data_ptr(), n, _stream(dev)), "so3_geodesic_f64")
        else:
            st = _stream(dev)
            ws = _workspace(dev, st) if n > _SMALL_BATCH else None          # one launch either way (above 1024 rows: the ticket finish)
            _check(_libh().so3_angle_error_v2_f64(a.data_ptr(), b_.data_ptr(), _ptr(rows), _ptr(sc), flag.data_ptr(), _ptr(ws),
                                                  _lib.RADIANS if radians else 0, n, st), "so3_angle_error_v2_f64")
    return rows, sc, flag


_WARNED = set()


def _warn_once(key: str, message: str) -> None:
    """One warning per process and key: a metric that silently drops a gradient is how a training run goes wrong without an error."""
    if key not in _WARNED:
        _WARNED.add(key)
        import warnings
        warnings.warn(message, RuntimeWarning, stacklevel=3)


def _wants_grad(*ts) -> bool:
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in ts)


def _metric_backward(ctx, grad, eps: float, radians: bool, f64_math: bool, divisor: float):
    """dR1, dR2 of a metric spelling for the upstream gradient `grad` (K4b, so3_angle_bwd_*): one launch writes the gradients the
    graph needs.  An upstream gradient that autograd EXPANDED from one element (the backward of .mean() / .sum() on the per-row
    result) travels as that one element; a 0-dim one (geodesic's own reductions) likewise -- a device pointer, no host sync."""
    _no_double_backward(grad)
    a, b_ = ctx.saved_tensors
    need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    (shape1, dtype1), (shape2, dtype2) = ctx.meta
    if not (need1 or need2):
        return None, None
    dev = a.device
    n = a.shape[0]
    data64 = a.dtype is torch.float64
    g_dtype = torch.float64 if (data64 or f64_math) else torch.float32
    if grad.dim() == 0 or (grad.dim() == 1 and n > 1 and grad.stride(0) == 0):
        g = grad.reshape(-1)[:1]
        scalar = True
    else:
        g = grad.reshape(-1)
        scalar = False
        if g.shape[0] != n:
            raise RuntimeError(f"metric backward: {g.shape[0]} upstream gradients for {n} rows")
    if g.dtype is not g_dtype:
        g = g.to(g_dtype)
    if not g.is_contiguous():
        g = g.contiguous()
    d1 = torch.empty_like(a) if need1 else None
    d2 = torch.empty_like(a) if need2 else None
    if n > 0:
        flags = (_lib.RADIANS if radians else 0) | (_lib.GRAD_SCALAR if scalar else 0)
        with _on_device(dev):
            if data64:
                _check(_libh().so3_angle_bwd_f64(a.data_ptr(), b_.data_ptr(), g.data_ptr(), divisor, eps, flags, _ptr(d1), _ptr(d2), n, _stream(dev)),
                       "so3_angle_bwd_f64")
            else:
                _check(_libh().so3_angle_bwd_f32(a.data_ptr(), b_.data_ptr(), g.data_ptr(), divisor, eps, flags | (_lib.F64_MATH if f64_math else 0),
                                                 _ptr(d1), _ptr(d2), n, _stream(dev)), "so3_angle_bwd_f32")
    if d1 is not None:
        d1 = (d1 if d1.dtype is dtype1 else d1.to(dtype1)).view(shape1)
    if d2 is not None:
        d2 = (d2 if d2.dtype is dtype2 else d2.to(dtype2)).view(shape2)
    return d1, d2


def _save_metric_inputs(ctx, r1, r2, f64: bool):
    """The (B,9) blocks the backward reads (the arguments themselves when they are contiguous float32 / float64 already)."""
    blocks = _f64_blocks if f64 else _f32_blocks
    a, b_ = blocks(r1), blocks(r2)
    ctx.save_for_backward(a, b_)
    ctx.meta = ((r1.shape, r1.dtype), (r2.shape, r2.dtype))
    return a, b_


class _AngleError(torch.autograd.Function):
    """angle_error as a graph node: the reference's function is plain differentiable tensor code (rotation_representation.py:230-242)."""

    @staticmethod
    def forward(ctx, r1, r2, check):
        f64 = _is_f64(r1, r2)
        a, b_ = _save_metric_inputs(ctx, r1, r2, f64)
        deg, _, flag = (_angle_call_f64 if f64 else _angle_call)(a, b_, True, False)
        if check and int(flag.item()) != 0:
            raise ValueError(_RANGE_MSG)
        return deg

    @staticmethod
    def backward(ctx, grad_deg):
        return (*_metric_backward(ctx, grad_deg, 0.0, False, True, 1.0), None)


def angle_error(t_R1: torch.Tensor, t_R2: torch.Tensor, check: bool = True) -> torch.Tensor:
    """Geodesic angle between rotations, float64 degrees, shape (B,).

    Raises ValueError("angle out of range, ...") when any cosine is outside [-1.1, 1.1], exactly as
    the reference does; that needs one device->host read (the reference's two `torch.any` cost two).
    `check=False` skips the read (and the raise) for benchmarking / graph capture.

    float64 arguments (the reference casts to float64 before the product, :232-233): K4 reads float32 data, so
    double tensors go to its float64 twin (so3_angle_error_v2_f64) instead of being rounded.

    Differentiable with respect to both arguments, like the reference's tensor code (K4b, so3_angle_bwd_*: the float64
    expression's gradient, rounded once to the argument's dtype; rows on the clamp get 0, as torch.clamp's backward gives).
    """
    if _wants_grad(t_R1, t_R2):
        _require_device(t_R1, t_R2)
        return _AngleError.apply(t_R1, t_R2, check)
    if _is_f64(t_R1, t_R2):
        deg, _, flag = _angle_call_f64(t_R1, t_R2, True, False)
    else:
        deg, _, flag = _angle_call(t_R1, t_R2, True, False)
    if check and int(flag.item()) != 0:
        raise ValueError(_RANGE_MSG)
    return deg


def angle_error_sum_count(t_R1: torch.Tensor, t_R2: torch.Tensor, check: bool = True) -> torch.Tensor:
    """Device tensor of two float64: (sum of angles in degrees, row count), reduced on the device.

    This pair is what one all-reduce sums across GPUs (poseestimation_amd.distributed); the
    per-row vector is never materialised."""
    if _wants_grad(t_R1, t_R2):
        _warn_once("angle_error_sum_count", "angle_error_sum_count is an evaluation call: the (sum, count) pair carries no gradient although an "
                                            "argument requires grad.  angle_error(...).sum() is the differentiable spelling.")
    _, sc, flag = (_angle_call_f64 if _is_f64(t_R1, t_R2) else _angle_call)(t_R1, t_R2, False, True)
    if check and int(flag.item()) != 0:
        raise ValueError(_RANGE_MSG)
    return sc


def head_angle_error(x: torch.Tensor, R_true: torch.Tensor, reduce: str = "none", check: bool = True, return_rotation: bool = False,
                     exact: bool = False):
    """Fused `angle_error(symmetric_orthogonalization(x), R_true)` (3D-Pose/main.py:60-62): one launch that reads
    x and R_true (72 B per row) and writes only what is asked for.

    reduce="none": (B,) float64 degrees;  reduce="mean": 0-dim float64 mean;  reduce="sum_count": the (sum, count)
    pair for a multi-GPU all-reduce.  return_rotation=True also returns R.  Not differentiable (evaluation path).

    Arithmetic.  reduce="none" gives, row for row, what `angle_error` gives on the materialised rotation (the reference's
    float64 expression, rotation_representation.py:232-241; equal to 1e-9 degrees).  The reduced forms of a batch above 1024
    rows evaluate the same expression -- trace, cosine, clamp, acos -- in float32 for every row whose cosine is at least
    5e-7 away from +-1 and in float64 for the rows inside that band (angles within 0.057 degrees of 0 or 180), and sum in
    float64: the result differs from `angle_error(...).mean()` by the float32 trace's round-off, at most 2e-7 / sin(theta) rad
    per row and without bias -- 3e-8 degrees on the mean of 1M Haar-distributed pairs, below 2e-6 degrees when every pair is
    0.3 degrees apart (the reference's own sensitivity to the 1e-7 of orthonormality defect its float32 inputs carry is larger).
    exact=True runs the float64 expression on every row (20 % slower at 1M rows).  The range check is the reference's."""
    if reduce not in ("none", "mean", "sum_count"):
        raise ValueError("reduce must be 'none', 'mean' or 'sum_count'")
    dev = _require_device(x, R_true)
    if _wants_grad(x, R_true):
        _warn_once("head_angle_error",
                   "head_angle_error is an evaluation call: its result carries no gradient although an argument requires grad.  "
                   "angle_error(symmetric_orthogonalization(x), R_true) and geodesic(...) are the differentiable spellings.")
    m = _head_input(x.detach())
    if m.dtype != torch.float32:
        m = m.float()
    t = _f32_blocks(R_true.detach())
    n = m.shape[0]
    if t.shape[0] != n:
        raise RuntimeError(f"head_angle_error: {n} predictions vs {t.shape[0]} targets")
    want_deg = reduce == "none"
    # the fused kernel handles whole 64-row units of 16-byte aligned arrays; anything else goes K1 -> R -> K4
    need_r = return_rotation or (n % 64 != 0) or (m.data_ptr() % 16 != 0) or (t.data_ptr() % 16 != 0)
    r = torch.empty((n, 3, 3), dtype=torch.float32, device=dev) if need_r else None
    deg = torch.empty((n,), dtype=torch.float64, device=dev) if want_deg else None
    with _on_device(dev):
        st = _stream(dev)
        sc, flag = _ZERO_POOL.take(dev, st)         # zero-filled slots: the kernel needs no init launch in front of it
        if want_deg:
            sc = None
        _check(_fn("so3_project_angle_error_v2_f32")(_ptr(m), _ptr(t), _ptr(r), _ptr(deg), _ptr(sc), _ptr(flag), None,
                                                      _lib.PREZEROED | (_lib.EXACT_F64 if exact else 0), n, st), "so3_project_angle_error_f32")
    if check and int(flag.item()) != 0:
        raise ValueError(_RANGE_MSG)
    out = deg if want_deg else (sc if reduce == "sum_count" else sc[0] / sc[1])
    return (out, r) if return_rotation else out


def _geodesic_rows(a, b_, dev):
    """K4' on (B,9) float32 / float64 blocks: radians, hard clamp."""
    if a.dtype is torch.float64:
        return _angle_call_f64(a, b_, True, False, geodesic=True)[0]
    n = a.shape[0]
    theta = torch.empty((n,), dtype=torch.float32, device=dev)
    with _on_device(dev):
        _check(_libh().so3_geodesic_f32(_ptr(a), _ptr(b_), _ptr(theta), n, _stream(dev)), "so3_geodesic_f32")
    return theta


class _GeodesicDistance(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m1, m2):
        a, b_ = _save_metric_inputs(ctx, m1, m2, _is_f64(m1, m2))
        return _geodesic_rows(a, b_, a.device)

    @staticmethod
    def backward(ctx, grad_theta):
        return _metric_backward(ctx, grad_theta, 0.0, True, False, 1.0)


def compute_geodesic_distance_from_two_matrices(m1: torch.Tensor, m2: torch.Tensor) -> torch.Tensor:
    """Geodesic distance in radians, tr(m1 m2^T), hard clamp to [-1, 1]; shape (B,); the arguments' dtype as the
    reference (rotation_representation.py:209-227): float32 through K4', float64 through so3_geodesic_f64.
    Differentiable with respect to both arguments (K4b; a row on the clamp gets 0, as torch.min / torch.max's backward gives)."""
    dev = _require_device(m1, m2)
    f64 = _is_f64(m1, m2)
    if not f64 and _as_blocks(m1).shape != _as_blocks(m2).shape:
        raise RuntimeError(f"geodesic: shape mismatch {tuple(m1.shape)} vs {tuple(m2.shape)}")
    if _wants_grad(m1, m2):
        return _GeodesicDistance.apply(m1, m2)
    if f64:
        return _angle_call_f64(m1, m2, True, False, geodesic=True)[0]
    return _geodesic_rows(_f32_blocks(m1), _f32_blocks(m2), dev)


def _geodesic_eps(a, b_, reduction, dev):
    """geodesic(...)'s launch on (B,9) blocks: float32 through so3_geodesic_eps_f32, float64 through its twin."""
    n = a.shape[0]
    dt = a.dtype
    f64 = dt is torch.float64
    theta = torch.empty((n,), dtype=dt, device=dev) if reduction == "none" else None
    acc = None if reduction == "none" else torch.empty((1,), dtype=torch.float64, device=dev)
    out = None if reduction == "none" else torch.empty((), dtype=dt, device=dev)
    with _on_device(dev):
        st = _stream(dev)
        if f64:
            _check(_libh().so3_geodesic_eps_f64(_ptr(a), _ptr(b_), _ptr(theta), _ptr(acc), _ptr(out), 1 if reduction == "mean" else 0, 1e-7, n, st),
                   "so3_geodesic_eps_f64")
        else:
            ws = _workspace(dev, st) if reduction != "none" else None           # the kernel's last workgroup writes the reduced value
            _check(_libh().so3_geodesic_eps_f32(_ptr(a), _ptr(b_), _ptr(theta), _ptr(acc), _ptr(out), 1 if reduction == "mean" else 0, 1e-7,
                                                _ptr(ws), n, st), "so3_geodesic_eps_f32")
    return theta if reduction == "none" else out


class _Geodesic(torch.autograd.Function):
    """geodesic(R1, R2, reduction) as a graph node: the use its eps was written for (point_cloud/main.py:64)."""

    @staticmethod
    def forward(ctx, r1, r2, reduction):
        a, b_ = _save_metric_inputs(ctx, r1, r2, _is_f64(r1, r2))
        ctx.divisor = float(a.shape[0]) if reduction == "mean" else 1.0
        return _geodesic_eps(a, b_, reduction, a.device)

    @staticmethod
    def backward(ctx, grad):
        if ctx.divisor == 0.0:                        # the mean of no rows: no rows to send a gradient to
            (s1, d1), (s2, d2) = ctx.meta
            a, _ = ctx.saved_tensors
            return (torch.zeros(s1, dtype=d1, device=a.device) if ctx.needs_input_grad[0] else None,
                    torch.zeros(s2, dtype=d2, device=a.device) if ctx.needs_input_grad[1] else None, None)
        return (*_metric_backward(ctx, grad, 1e-7, True, False, ctx.divisor), None)


def geodesic(R1: torch.Tensor, R2: torch.Tensor, reduction: str = "mean"):
    """point_cloud/main.py:61-73: acos(clamp((tr(R1 R2^T) - 1)/2, -1 + 1e-7, 1 - 1e-7)), radians, in the arguments' dtype (float32
    on the hot path; float64 arguments go to the float64 twin);
    reduction "none" -> (B,), "mean" / "sum" -> 0-dim; any other string returns None, as the reference's if-chain does.
    Angles and their sum leave one launch (float64 accumulation; the reference's float32 .mean() agrees to its own round-off).
    Differentiable with respect to both arguments (K4b, so3_angle_bwd_*): one launch for dR1 and dR2, the reduction's 1/B folded in."""
    dev = _require_device(R1, R2)
    f64 = _is_f64(R1, R2)
    blocks = _f64_blocks if f64 else _f32_blocks
    if _as_blocks(R1).shape != _as_blocks(R2).shape:
        raise RuntimeError(f"geodesic: shape mismatch {tuple(R1.shape)} vs {tuple(R2.shape)}")
    if reduction not in ("none", "mean", "sum"):
        return None
    if _wants_grad(R1, R2):
        return _Geodesic.apply(R1, R2, reduction)
    return _geodesic_eps(blocks(R1), blocks(R2), reduction, dev)


# --------------------------------------------------------------------------------------------
# K3: loss
# --------------------------------------------------------------------------------------------
class _LossFrobenius(torch.autograd.Function):
    @staticmethod
    def forward(ctx, r_pred, r_true):
        dev = _require_device(r_pred, r_true)
        f64 = r_pred.dtype is torch.float64 or r_true.dtype is torch.float64      # torch's promotion: the loss is float64 then
        blocks = _f64_blocks if f64 else _f32_blocks
        p, t = blocks(r_pred), blocks(r_true)
        if p.shape != t.shape:
            raise RuntimeError(f"loss_frobenius: shape mismatch {tuple(r_pred.shape)} vs {tuple(r_true.shape)}")
        b = p.shape[0]
        need_grad = r_pred.requires_grad or r_true.requires_grad
        g = torch.empty_like(p) if need_grad else None
        loss_sum = torch.empty((1,), dtype=torch.float64, device=dev)
        loss = torch.empty((), dtype=torch.float64 if f64 else torch.float32, device=dev)       # the mean is written by the kernel
        with _on_device(dev):
            st = _stream(dev)
            if f64:
                ws = _workspace(dev, st) if b > _SMALL_BATCH else None
                _check(_libh().so3_frob_loss_v2_f64(p.data_ptr(), t.data_ptr(), _ptr(g), loss_sum.data_ptr(), loss.data_ptr(), _ptr(ws), 0, b, st),
                       "so3_frob_loss_v2_f64")
            else:
                ws = _workspace(dev, st) if b > _SMALL_BATCH else None
                _check(_fn("so3_frob_loss_v2_f32")(p.data_ptr(), t.data_ptr(), _ptr(g), loss_sum.data_ptr(), loss.data_ptr(), _ptr(ws), 0, b, st),
                       "so3_frob_loss_f32")
        ctx.g = g
        ctx.shapes = (r_pred.shape, r_true.shape, r_pred.dtype, r_true.dtype)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_loss):
        if ctx.g is None:
            return None, None
        sp, st, dp, dt = ctx.shapes
        g = ctx.g * grad_loss
        gp = g.to(dp).view(sp) if ctx.needs_input_grad[0] else None
        gt = (-g).to(dt).view(st) if ctx.needs_input_grad[1] else None
        return gp, gt


def loss_frobenius(R_pred: torch.Tensor, R_true: torch.Tensor) -> torch.Tensor:
    """mean_b ||R_true - R_pred||_F (not squared), differentiable w.r.t. both arguments.

    Stand-alone form for callers that already hold R_pred (one kernel for the loss and its gradient).
    A training step should use `frobenius_head`, which fuses head, loss and backward into one launch.

    Returns the arguments' dtype as the reference (3D-Pose/loss.py:7-11): float32 through K3'; if either argument is
    float64 (e.g. the float64 head's output) its float64 twin, so3_frob_loss_v2_f64."""
    node = _node()
    if node is not None and type(R_pred) is torch.Tensor and type(R_true) is torch.Tensor and R_pred.is_cuda:
        dev = R_pred.device
        st = _stream(dev)
        ws = 0
        if R_pred.numel() > 9 * _SMALL_BATCH:
            w = _workspace(dev, st)
            ws = w.data_ptr() if w is not None else 0
        loss = node.loss_frobenius(R_pred, R_true, st, ws)                   # the C++ node; None for what it does not cover
        if loss is not None:
            return loss
    return _LossFrobenius.apply(R_pred, R_true)


class _FrobeniusHead(torch.autograd.Function):
    """One differentiable output (the loss); the rotation, which carries no gradient, leaves through `box` instead of being a
    second output that autograd would have to wrap, mark and track."""

    @staticmethod
    def forward(ctx, x, r_true, want_r, box):
        dev = x.device if (x.is_cuda and r_true.is_cuda and x.device == r_true.device) else _require_device(x, r_true)
        m = _head_input(x)
        b = m.shape[0]
        if r_true.dtype is torch.float32 and r_true.is_contiguous() and r_true.numel() == 9 * b:
            t = r_true                                               # (B,3,3) or (B,9) as stored: only its address is needed
        else:
            t = _f32_blocks(r_true)
            if t.shape[0] != b:
                raise RuntimeError(f"frobenius_head: {b} predictions vs {t.shape[0]} targets")
        need_grad = x.requires_grad
        # d loss / d R_true = -(R - R_true) / (B ||R - R_true||_F), the loss being differentiable in both arguments
        # (3D-Pose/loss.py:7-11): it is rebuilt in backward from R and R_true (K3'), so R is kept whenever it is asked for
        ctx.true_grad = r_true.requires_grad
        want_r_user = want_r                      # the caller gets R (and may write into it)
        want_r = want_r or ctx.true_grad
        r = torch.empty((b, 3, 3), dtype=torch.float32, device=dev) if want_r else None
        dm = torch.empty_like(m) if need_grad else None
        loss_sum = torch.empty((1,), dtype=torch.float64, device=dev) if b > _SMALL_BATCH or b == 0 else None
        loss = torch.empty((), dtype=torch.float32, device=dev)      # the kernel writes the float32 mean itself: no launch of ours
        fn = _fn("so3_frob_fwd_bwd_v2_bf16" if m.dtype is torch.bfloat16 else "so3_frob_fwd_bwd_v2_f32")
        with _on_device(dev):
            st = _stream(dev)
            ws = _workspace(dev, st) if b > _SMALL_BATCH else None
            _check(fn(m.data_ptr(), t.data_ptr(), _ptr(r), _ptr(dm), _ptr(loss_sum), loss.data_ptr(), _ptr(ws), 0, b, st), "so3_frob_fwd_bwd")
        ctx.dm = dm
        ctx.in_shape = x.shape
        ctx.in_dtype = x.dtype
        if ctx.true_grad:
            # the target goes through save_for_backward (an in-place edit between forward and backward then raises instead of
            # yielding a silently wrong gradient); the rotation handed to the caller leaves autograd through `box`, so backward
            # keeps a private copy of it
            ctx.save_for_backward(t)
            ctx.rt = (r.clone() if want_r_user else r, r_true.shape, r_true.dtype)
        box.append(r)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        _no_double_backward(grad_loss)
        dm = ctx.dm
        gx = gt = None
        if dm is not None and ctx.needs_input_grad[0]:
            # out of place: a second backward over the same graph (retain_graph, several losses) must see the stored gradient
            # unscaled, and the tensor handed out must not alias it.  One launch of ours (so3_scale_*) instead of torch's
            # float() / mul / to(bfloat16) chain: the upstream factor is a 0-dim float32 device tensor.
            if grad_loss.dtype is torch.float32 and grad_loss.is_cuda and dm.dtype is ctx.in_dtype:
                dev = dm.device
                gx = torch.empty_like(dm)
                fn = _fn("so3_scale_bf16" if dm.dtype is torch.bfloat16 else "so3_scale_f32")
                with _on_device(dev):
                    _check(fn(dm.data_ptr(), grad_loss.data_ptr(), gx.data_ptr(), dm.numel(), _stream(dev)), "so3_scale")
                gx = gx.view(ctx.in_shape)
            elif dm.dtype == ctx.in_dtype:
                gx = (dm * grad_loss).view(ctx.in_shape)
            else:
                gx = (dm.float() * grad_loss).to(ctx.in_dtype).view(ctx.in_shape)
        if ctx.true_grad and ctx.needs_input_grad[1]:
            r, shape, dtype = ctx.rt
            (t,) = ctx.saved_tensors
            dev = t.device
            b = t.shape[0]
            g = torch.empty_like(t)                              # d(mean loss)/dR_pred; the target's gradient is its negative
            scratch = torch.empty((1,), dtype=torch.float64, device=dev)
            with _on_device(dev):
                _check(_libh().so3_frob_loss_v2_f32(_ptr(r), _ptr(t), _ptr(g), _ptr(scratch), None, None, 0, b, _stream(dev)), "so3_frob_loss_f32")
            gt = (g * (-grad_loss)).to(dtype).view(shape)
        return gx, gt, None, None


def frobenius_head(x: torch.Tensor, R_true: torch.Tensor, return_rotation: bool = True):
    """Fused `out = symmetric_orthogonalization(x); loss = loss_frobenius(R_true, out)`.

    One kernel computes R, the loss and d(loss)/dx; `loss.backward()` then only scales the stored
    gradient.  Returns (loss, R) -- R is detached (use it for metrics) -- or loss alone.
    float64 x: the fused kernel is float32 / bfloat16 only, so the float64 head and the float64 loss are composed
    (same values and dtypes as the reference's two calls).
    """
    if x.dtype is torch.float64:
        r64 = symmetric_orthogonalization(x)
        loss64 = loss_frobenius(R_true.to(torch.float64), r64)
        return (loss64, r64.detach()) if return_rotation else loss64
    node = _node()
    if node is not None and type(x) is torch.Tensor and type(R_true) is torch.Tensor and x.is_cuda and x.dim() >= 2:
        # the C++ node: same launches, no interpreter inside forward / backward; it declines (None) what it does not cover
        dev = x.device
        st = _stream(dev)
        ws = 0
        if x.shape[0] > _SMALL_BATCH:
            w = _workspace(dev, st)
            ws = w.data_ptr() if w is not None else 0
        res = node.frobenius_head(x, R_true, return_rotation, st, ws)
        if res is not None:
            return res if return_rotation else res[0]
    box = []
    loss = _FrobeniusHead.apply(x, R_true, return_rotation, box)
    return (loss, box[0]) if return_rotation else loss


class FrobeniusHeadStep:
    """The tail of a training step -- head, Frobenius loss and d(loss)/dx (3D-Pose/main.py:60,85,90) -- for a FIXED batch
    shape, recorded once into a hipGraph and replayed: config #4 (B = 512) is launch-bound, and through autograd the
    Python and engine bookkeeping around the 5-us kernel costs twenty times the kernel.

        step = FrobeniusHeadStep(512, dtype=torch.bfloat16, device="cuda:0")
        step.x.copy_(network_output); step.r_true.copy_(targets)      # or write into them directly
        loss, dx, r = step()                                          # one graph replay; tensors are reused between calls
        network_output.backward(dx)                                   # continue into the backbone

    `x`, `r_true` are the static inputs; `loss` (0-dim float32 mean), `dx` (like x) and `r` (B,3,3) are overwritten by
    every call.  The C ABI is enqueue-only with caller-owned buffers, which is what makes it capturable."""

    def __init__(self, batch: int, dtype: torch.dtype = torch.float32, device="cuda", return_rotation: bool = True):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("FrobeniusHeadStep needs a HIP device (there is no CPU fallback)")
        if dtype not in (torch.float32, torch.bfloat16):
            raise TypeError("FrobeniusHeadStep: float32 or bfloat16 input")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.batch = int(batch)
        self.x = torch.zeros((self.batch, 9), dtype=dtype, device=dev)
        self.r_true = torch.eye(3, device=dev).repeat(self.batch, 1, 1)
        self.dx = torch.empty_like(self.x)
        self.r = torch.empty((self.batch, 3, 3), dtype=torch.float32, device=dev) if return_rotation else None
        self._sum = torch.empty((1,), dtype=torch.float64, device=dev)
        self.loss = torch.empty((), dtype=torch.float32, device=dev)
        lib = _libh()
        fn = lib.so3_frob_fwd_bwd_v2_bf16 if dtype == torch.bfloat16 else lib.so3_frob_fwd_bwd_v2_f32
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))

        def record():                 # the float32 mean is written by the kernel(s); no workspace inside a graph
            _check(fn(_ptr(self.x), _ptr(self.r_true), _ptr(self.r), _ptr(self.dx), _ptr(self._sum), _ptr(self.loss), None, 0, self.batch,
                      side.cuda_stream), "so3_frob_fwd_bwd")

        with torch.cuda.device(dev), torch.cuda.stream(side):
            record()                                                                 # warm-up outside the capture
            side.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph, stream=side, capture_error_mode="thread_local"):
                record()
        torch.cuda.current_stream(dev).wait_stream(side)

    def __call__(self):
        self._graph.replay()
        return self.loss, self.dx, self.r


# --------------------------------------------------------------------------------------------
# K5: Kabsch
# --------------------------------------------------------------------------------------------
def kabsch_rotation(P: torch.Tensor, Q: torch.Tensor, return_h: bool = False):
    """R_b = argmin_R sum_i |R p_bi - q_bi|^2 over SO(3) = proj(sum_i q_bi p_bi^T).

    P, Q: (B, N, 3) float32 clouds as in point_cloud/main.py:171-181 (q = R p, no translation).
    Equivalent to symmetric_orthogonalization(torch.bmm(Q.transpose(1, 2), P)) in one launch."""
    dev = _require_device(P, Q)
    if P.dim() != 3 or P.shape[-1] != 3 or P.shape != Q.shape:
        raise RuntimeError(f"kabsch_rotation: expected two (B, N, 3) tensors, got {tuple(P.shape)} and {tuple(Q.shape)}")
    p = P.detach().contiguous().float()
    q = Q.detach().contiguous().float()
    b, n, _ = p.shape
    r = torch.empty((b, 3, 3), dtype=torch.float32, device=dev)
    h = torch.empty((b, 3, 3), dtype=torch.float32, device=dev) if return_h else None
    with _on_device(dev):
        _check(_libh().so3_kabsch_f32(_ptr(p), _ptr(q), _ptr(r), _ptr(h), b, n, _stream(dev)), "so3_kabsch_f32")
    return (r, h) if return_h else r


# --------------------------------------------------------------------------------------------
# row a7: the cloud side of the point-cloud path
# --------------------------------------------------------------------------------------------
def rotate_point_clouds(pc: torch.Tensor, R: torch.Tensor, transposed: bool = False) -> torch.Tensor:
    """q_bi = R_b p_bi for every point of every cloud: the pairing rule of point_cloud/main.py:173-181 (expand the
    rotation to all points, bmm, view) in one launch.  pc: (B,N,3), R: (B,3,3).  Returns (B,N,3), or with
    `transposed` the contiguous (B,3,N) tensor the reference obtains from `.transpose(1, 2)` at :183."""
    dev = _require_device(pc, R)
    if pc.dim() != 3 or pc.shape[-1] != 3 or R.numel() != pc.shape[0] * 9:
        raise RuntimeError(f"rotate_point_clouds: expected (B, N, 3) and (B, 3, 3), got {tuple(pc.shape)} and {tuple(R.shape)}")
    p = pc.detach().contiguous().float()
    r = R.detach().reshape(-1, 9).contiguous().float()
    b, n, _ = p.shape
    out = torch.empty((b, 3, n) if transposed else (b, n, 3), dtype=torch.float32, device=dev)
    with _on_device(dev):
        _check(_libh().so3_rotate_clouds_f32(_ptr(p), _ptr(r), _ptr(out), 1 if transposed else 0, b, n, _stream(dev)), "so3_rotate_clouds_f32")
    return out


def pc_normalize(pc: torch.Tensor):
    """Centre a cloud on its bounding box and scale by the box diagonal; point_cloud/prepare.py:51-56.
    pc: (N,3) as the reference takes it, or a (B,N,3) batch.  Returns (pc, centroid, scale) like the reference
    (centroid (3,) / (B,3); scale 0-dim / (B,)), float32 on the device."""
    dev = _require_device(pc)
    single = pc.dim() == 2
    p = (pc.unsqueeze(0) if single else pc).detach().contiguous().float()
    if p.dim() != 3 or p.shape[-1] != 3 or p.shape[1] < 1:
        raise RuntimeError(f"pc_normalize: expected (N, 3) or (B, N, 3) with N >= 1, got {tuple(pc.shape)}")
    b, n, _ = p.shape
    out, cen, sc = torch.empty_like(p), torch.empty((b, 3), dtype=torch.float32, device=dev), torch.empty((b,), dtype=torch.float32, device=dev)
    with _on_device(dev):
        _check(_libh().so3_pc_normalize_f32(_ptr(p), _ptr(out), _ptr(cen), _ptr(sc), b, n, _stream(dev)), "so3_pc_normalize_f32")
    return (out[0], cen[0], sc[0]) if single else (out, cen, sc)


# --------------------------------------------------------------------------------------------
# next row f4: on-device pair synthesis for Kabsch
# --------------------------------------------------------------------------------------------
def get_sampled_rotation_matrices_by_axisAngle(batch: int, device="cuda", generator: torch.Generator = None) -> torch.Tensor:
    """Random rotations by the reference's recipe (point_cloud/prepare.py:21-49): theta ~ U(-pi, pi), axis =
    normalised N(0, I), quaternion (cos theta, axis sin theta).  torch draws the random numbers; the quaternion ->
    matrix arithmetic runs in the HIP library."""
    dev = torch.device(device)
    theta = (torch.rand(batch, device=dev, generator=generator) * 2 - 1) * torch.pi
    axis = torch.randn(batch, 3, device=dev, generator=generator)
    return rotations_from_axis_angle_draws(theta, axis)


def rotations_from_axis_angle_draws(theta: torch.Tensor, axis: torch.Tensor) -> torch.Tensor:
    dev = _require_device(theta, axis)
    t = theta.detach().reshape(-1).contiguous().float()
    a = axis.detach().reshape(-1, 3).contiguous().float()
    if a.shape[0] != t.shape[0]:
        raise RuntimeError("rotations_from_axis_angle_draws: theta (B,) and axis (B,3) disagree")
    r = torch.empty((t.shape[0], 3, 3), dtype=torch.float32, device=dev)
    with _on_device(dev):
        _check(_libh().so3_rotations_axis_angle_f32(_ptr(t), _ptr(a), _ptr(r), t.shape[0], _stream(dev)), "so3_rotations_axis_angle_f32")
    return r


def kabsch_rotation_synthetic(P: torch.Tensor, R_gt: torch.Tensor, sigma: float = 0.0, seed: int = 0, return_h: bool = False):
    """Kabsch with the second cloud synthesised in the kernel: q = R_gt p + sigma * n(seed, cloud, point).
    Only P is read from HBM (config #3 with half the traffic)."""
    dev = _require_device(P, R_gt)
    if P.dim() != 3 or P.shape[-1] != 3 or R_gt.shape[0] != P.shape[0]:
        raise RuntimeError("kabsch_rotation_synthetic: expected P (B,N,3) and R_gt (B,3,3)")
    p = P.detach().contiguous().float()
    g = R_gt.detach().reshape(-1, 9).contiguous().float()
    b, n, _ = p.shape
    r = torch.empty((b, 3, 3), dtype=torch.float32, device=dev)
    h = torch.empty((b, 3, 3), dtype=torch.float32, device=dev) if return_h else None
    with _on_device(dev):
        _check(_libh().so3_kabsch_synth_f32(_ptr(p), _ptr(g), float(sigma), int(seed) & 0xFFFFFFFF, _ptr(r), _ptr(h), b, n, _stream(dev)),
                   "so3_kabsch_synth_f32")
    return (r, h) if return_h else r


# --------------------------------------------------------------------------------------------
# next row f1: the SE(3) pose update of the iterative refiner
# --------------------------------------------------------------------------------------------
def get_scene_parameters():
    """Focal lengths in pixels, as Iterative/utility.py:73-88: 50 mm lens, 36 mm sensor, 320 px."""
    sw, img_res, flen = 36, 320, 50
    fx = fy = flen / (sw / img_res)
    return fx, fy


class _Se3Update(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model_output, t_init, fx, fy):
        dev = _require_device(model_output, t_init)
        if model_output.dim() != 2 or model_output.shape[1] < 12 or tuple(t_init.shape[1:]) != (4, 4):
            raise RuntimeError("calculate_T_pred expects model_output (B, >=12) and T_init (B, 4, 4)")
        o = model_output.detach()[:, :12].contiguous().float()
        t = t_init.detach().reshape(-1, 16).contiguous().float()
        b = o.shape[0]
        tp = torch.empty((b, 4, 4), dtype=torch.float32, device=dev)
        with _on_device(dev):
            _check(_libh().so3_se3_update_f32(_ptr(o), _ptr(t), _ptr(tp), fx, fy, b, _stream(dev)), "so3_se3_update_f32")
        ctx.save_for_backward(o, t)
        ctx.meta = (model_output.shape, model_output.dtype, fx, fy)
        return tp

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_t):
        o, t = ctx.saved_tensors
        shape, dtype, fx, fy = ctx.meta
        dev = o.device
        g = grad_t.reshape(-1, 16).contiguous().float()
        d = torch.empty_like(o)
        with _on_device(dev):
            _check(_libh().so3_se3_update_bwd_f32(_ptr(o), _ptr(t), _ptr(g), _ptr(d), fx, fy, o.shape[0], _stream(dev)), "so3_se3_update_bwd_f32")
        full = torch.zeros(shape, dtype=torch.float32, device=dev)
        full[:, :12] = d
        return full.to(dtype), None, None, None


def calculate_T_pred(model_output: torch.Tensor, T_init: torch.Tensor, device=None, rot_repr: str = "SVD") -> torch.Tensor:
    """SE(3) update of the iterative refiner (Iterative/utility.py:90-128), one fused launch.

    model_output: (B,12) = 9 numbers for the SVD head + (vx, vy, vz); T_init: (B,4,4).  Returns T_pred (B,4,4)
    float32, differentiable w.r.t. model_output (T_init is a constant, as the reference's loop detaches it).
    `device` is accepted for signature compatibility and ignored (the result lives where the inputs do), and so is
    `rot_repr`: the reference never reads it -- its body always runs the SVD head on the first nine outputs
    (Iterative/utility.py:105), whatever the string says."""
    fx, fy = get_scene_parameters()
    return _Se3Update.apply(model_output, T_init, float(fx), float(fy))


# --------------------------------------------------------------------------------------------
# next row f6: the ADD-L1 losses on calculate_T_pred's output (Iterative/loss.py)
# --------------------------------------------------------------------------------------------
def _add_l1_args(t_gt, t_pred, points):
    bsz = len(t_gt)
    assert t_pred.shape == (bsz, 4, 4) and t_gt.shape == (bsz, 4, 4)            # reference: Iterative/loss.py:18
    assert points.dim() == 3 and points.shape[-1] == 3                          # :19
    assert points.shape[0] == bsz                                               # transform_pts, :58
    dev = _require_device(t_pred)
    _require_device(t_gt)
    _require_device(points)
    if points.shape[1] < 1:
        raise RuntimeError("ADD-L1: at least one model point per sample is needed")
    prep = lambda t: t.detach().contiguous().float()
    return dev, bsz, int(points.shape[1]), prep(t_gt), prep(t_pred), prep(points)


class _AddL1(torch.autograd.Function):
    """compute_ADD_L1_loss and its gradient w.r.t. the predicted pose, one launch (so3_add_l1_f32)."""

    @staticmethod
    def forward(ctx, t_gt, t_pred, points, use_batch_mean):
        dev, b, n, tg, tp, pts = _add_l1_args(t_gt, t_pred, points)
        want_grad = t_pred.requires_grad
        dt = torch.empty((b, 4, 4), dtype=torch.float32, device=dev) if want_grad else None
        dists = None if use_batch_mean else torch.empty((b,), dtype=torch.float32, device=dev)
        loss_sum = torch.empty((1,), dtype=torch.float64, device=dev) if use_batch_mean else None
        scale = 1.0 / max(b, 1) if use_batch_mean else 1.0
        with _on_device(dev):
            _check(_libh().so3_add_l1_f32(_ptr(tg), _ptr(tp), _ptr(pts), _ptr(dists), _ptr(loss_sum), _ptr(dt), scale, b, n,
                                                  _stream(dev)), "so3_add_l1_f32")
        ctx.dt, ctx.per_sample, ctx.in_dtype = dt, not use_batch_mean, t_pred.dtype
        if use_batch_mean:
            return loss_sum.to(torch.float32).mul_(1.0 / max(b, 1)).squeeze(0)
        return dists

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        if ctx.dt is None:
            return None, None, None, None
        g = grad_out.reshape(-1, 1, 1) if ctx.per_sample else grad_out
        return None, (ctx.dt * g).to(ctx.in_dtype), None, None


class _AddL1Disentangled(torch.autograd.Function):
    """compute_disentangled_ADD_L1_loss and its gradient, one launch (so3_add_l1_disentangled_f32)."""

    @staticmethod
    def forward(ctx, t_pred, t_gt, points):
        dev, b, n, tg, tp, pts = _add_l1_args(t_gt, t_pred, points)
        dt = torch.empty((b, 4, 4), dtype=torch.float32, device=dev) if t_pred.requires_grad else None
        loss_sum = torch.empty((3,), dtype=torch.float64, device=dev)
        with _on_device(dev):
            _check(_libh().so3_add_l1_disentangled_f32(_ptr(tp), _ptr(tg), _ptr(pts), _ptr(loss_sum), _ptr(dt), 1.0 / max(b, 1),
                                                               b, n, _stream(dev)), "so3_add_l1_disentangled_f32")
        ctx.dt, ctx.in_dtype = dt, t_pred.dtype
        return loss_sum.sum().to(torch.float32).mul_(1.0 / max(b, 1))

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        if ctx.dt is None:
            return None, None, None
        return (ctx.dt * grad_out).to(ctx.in_dtype), None, None


def compute_ADD_L1_loss(TCO_gt: torch.Tensor, TCO_pred: torch.Tensor, points: torch.Tensor, use_batch_mean: bool = True) -> torch.Tensor:
    """mean |T_gt p - T_pred p| over points and coordinates (and the batch); Iterative/loss.py:10-26.
    Differentiable w.r.t. TCO_pred (the ground-truth pose and the model points are constants in the reference's loops)."""
    return _AddL1.apply(TCO_gt, TCO_pred, points, bool(use_batch_mean))


def compute_disentangled_ADD_L1_loss(T_CO_pred: torch.Tensor, T_CO_gt: torch.Tensor, points: torch.Tensor) -> torch.Tensor:
    """rotation + xy-translation + depth ADD-L1 terms; Iterative/loss.py:29-48, called right after calculate_T_pred
    (Iterative/main.py:94-95).  Differentiable w.r.t. T_CO_pred."""
    return _AddL1Disentangled.apply(T_CO_pred, T_CO_gt, points)


# --------------------------------------------------------------------------------------------
# next row f3: per-class evaluation statistics
# --------------------------------------------------------------------------------------------
STAT_FIELDS = ("count", "mean", "std", "max", "median", "acc30", "acc15", "acc7.5")
_STAT_WORKSPACES = {}


def angle_error_statistics(angles: torch.Tensor, class_ids: torch.Tensor = None, num_classes: int = 1) -> dict:
    """Device-side replacement of the numpy block in 3D-Pose/test_per_class.py:174-216.

    angles: (B,) degrees (e.g. `angle_error(...)`); class_ids: optional (B,) integer ids in [0, num_classes).
    Returns {"count","mean","std","max","median","acc30","acc15","acc7.5"} -> (num_classes,) float64 tensors;
    median is exact (radix select), std is numpy's population std, acc* are (x < t).sum()/len(x)."""
    dev = _require_device(angles)
    a = angles.detach().reshape(-1).contiguous().double()
    c = None
    if class_ids is not None:
        _require_device(class_ids)
        c = class_ids.detach().reshape(-1).contiguous().to(torch.int32)
        if c.numel() != a.numel():
            raise RuntimeError("angle_error_statistics: angles and class_ids differ in length")
    lib = _libh()
    stats = torch.empty((num_classes, len(STAT_FIELDS)), dtype=torch.float64, device=dev)
    with _on_device(dev):
        st = _stream(dev)
        # the statistics' workspace of (device, stream): zero-filled once, every call leaves it usable (include/so3proj.h) -- 10 MB kept
        # per stream that ever asked; a fresh zero-filled one while the stream is being captured (a replay may run beside eager calls)
        key = (dev.index, st)
        work = None if _capturing(dev) else _STAT_WORKSPACES.get(key)
        if work is None:
            work = torch.zeros((lib.so3_angle_stats_workspace_bytes(),), dtype=torch.uint8, device=dev)
            if not _capturing(dev):
                _STAT_WORKSPACES[key] = work
        code = lib.so3_angle_stats(_ptr(a), _ptr(c), num_classes, _ptr(stats), _ptr(work), a.numel(), st)
        if code != 0:
            # include/so3proj.h: a workspace is to be re-zeroed after a call that returned an error -- a refused call may have
            # enqueued its first launch: the cached one is dropped, the next call starts from a fresh zero-filled one
            _STAT_WORKSPACES.pop(key, None)
            _check(code, "so3_angle_stats")
    return {name: stats[:, i] for i, name in enumerate(STAT_FIELDS)}


# --------------------------------------------------------------------------------------------
# next row f2: the 6D Gram-Schmidt head
# --------------------------------------------------------------------------------------------
class _Ortho6d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, poses):
        dev = _require_device(poses)
        if poses.shape[-1] != 6:
            raise AssertionError("compute_rotation_matrix_from_ortho6d expects (..., 6) poses")   # reference: assert, :28
        x = poses.detach().reshape(-1, 6).contiguous().float()
        b = x.shape[0]
        r = torch.empty((b, 3, 3), dtype=torch.float32, device=dev)
        with _on_device(dev):
            _check(_libh().so3_ortho6d_fwd_f32(_ptr(x), _ptr(r), b, _stream(dev)), "so3_ortho6d_fwd_f32")
        ctx.save_for_backward(x)
        ctx.in_shape, ctx.in_dtype = poses.shape, poses.dtype
        return r.view(*poses.shape[:-1], 3, 3)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_r):
        (x,) = ctx.saved_tensors
        dev = x.device
        g = grad_r.reshape(-1, 9).contiguous().float()
        dx = torch.empty_like(x)
        with _on_device(dev):
            _check(_libh().so3_ortho6d_bwd_f32(_ptr(x), _ptr(g), _ptr(dx), x.shape[0], _stream(dev)), "so3_ortho6d_bwd_f32")
        return dx.to(ctx.in_dtype).view(ctx.in_shape)


def compute_rotation_matrix_from_ortho6d(poses: torch.Tensor) -> torch.Tensor:
    """6D (two 3-vectors) -> rotation by Gram-Schmidt, columns (x, y, z); rotation_representation.py:21-36.
    poses: (..., 6); returns (..., 3, 3) float32, differentiable."""
    r = _row_head("ortho6d", 6, poses)
    return r if r is not None else _Ortho6d.apply(poses)


# --------------------------------------------------------------------------------------------
# next row f5: the other heads of the reference's dispatch tables
# --------------------------------------------------------------------------------------------
def _make_head(symbol: str, width: int):
    """autograd.Function over so3_<symbol>_fwd_f32 / _bwd_f32 for a (B, width) -> (B, 3, 3) head."""

    class _Head(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x_in):
            dev = _require_device(x_in)
            x = x_in.detach().contiguous().float()
            b = x.shape[0]
            r = torch.empty((b, 3, 3), dtype=torch.float32, device=dev)
            with _on_device(dev):
                _check(getattr(_lib.load(), "so3_%s_fwd_f32" % symbol)(_ptr(x), _ptr(r), b, _stream(dev)), "so3_%s_fwd_f32" % symbol)
            ctx.save_for_backward(x)
            ctx.in_dtype = x_in.dtype
            return r

        @staticmethod
        @once_differentiable
        def backward(ctx, grad_r):
            (x,) = ctx.saved_tensors
            dev = x.device
            g = grad_r.reshape(-1, 9).contiguous().float()
            dx = torch.empty_like(x)
            with _on_device(dev):
                _check(getattr(_lib.load(), "so3_%s_bwd_f32" % symbol)(_ptr(x), _ptr(g), _ptr(dx), x.shape[0], _stream(dev)), "so3_%s_bwd_f32" % symbol)
            return dx.to(ctx.in_dtype)

    _Head.__name__ = "_Head_" + symbol
    return _Head


_Quat, _Euler, _Ortho5d, _ExpMap = (_make_head(sym, n) for sym, n in (("quat", 4), ("euler", 3), ("ortho5d", 5), ("expmap", 3)))


def _batch_of(x: torch.Tensor, width: int, name: str) -> None:
    if x.dim() != 2 or x.shape[1] != width:          # the reference indexes [:, k] and .view(batch, 1): 2-D input only
        raise RuntimeError("%s expects a (batch, %d) tensor, got %s" % (name, width, tuple(x.shape)))


def compute_rotation_matrix_from_quaternion(quaternion: torch.Tensor) -> torch.Tensor:
    """(B,4) quaternion (w,x,y,z), normalised with max(|q|, 1e-8) -> (B,3,3); rotation_representation.py:137-171."""
    _batch_of(quaternion, 4, "compute_rotation_matrix_from_quaternion")
    r = _row_head("quat", 4, quaternion)
    return r if r is not None else _Quat.apply(quaternion)


def compute_rotation_matrix_from_euler(euler: torch.Tensor) -> torch.Tensor:
    """(B,3) Euler angles -> (B,3,3) in the reference's convention (c2,s2 from column 2, c3,s3 from column 1);
    rotation_representation.py:92-113."""
    _batch_of(euler, 3, "compute_rotation_matrix_from_euler")
    r = _row_head("euler", 3, euler)
    return r if r is not None else _Euler.apply(euler)


def compute_rotation_matrix_from_ortho5d(a: torch.Tensor) -> torch.Tensor:
    """(B,5) -> (B,3,3): stereographic un-projection of a[:,2:5] to a unit 4-vector, then the 6D head;
    rotation_representation.py:118-134."""
    _batch_of(a, 5, "compute_rotation_matrix_from_ortho5d")
    r = _row_head("ortho5d", 5, a)
    return r if r is not None else _Ortho5d.apply(a)


def so3_exp_map(log_rot: torch.Tensor, eps: float = 0.0001) -> torch.Tensor:
    """so(3) exponential map with PyTorch3D's clamp; rotation_representation.py:245-275."""
    if log_rot.dim() != 2 or log_rot.shape[1] != 3:
        raise ValueError("Input tensor shape has to be Nx3.")                   # reference: :255-256
    if eps != 0.0001:
        raise NotImplementedError("so3_exp_map: the kernel is built for the reference's only eps, 1e-4")
    r = _row_head("expmap", 3, log_rot)
    return r if r is not None else _ExpMap.apply(log_rot)


def vec_3d_to_SO3(x: torch.Tensor) -> torch.Tensor:
    """transform_output['3D']: (B,3) -> (B,3,3) through so3_exp_map; rotation_representation.py:309-321."""
    assert x.dim() == 2 and x.shape[1] == 3                                      # reference: assert, :316
    return so3_exp_map(x)


# Head dispatch tables, keyed like the reference's.  `transform_output` is rotation_representation.py:323-324;
# `head_functions` / `head_dimensions` are Model.func / Model.dimension of Comparison/models.py:18-19 and
# point_cloud/model_fetch.py:153-154 ("Direct" is a reshape there and has no function), with 3D-Pose/main.py:46's
# lower-case "quat" as an alias.  The SVD head is the scope of SURVEY.md section 8; the rest are its next rows f2, f5.
transform_output = {"SVD": (9, symmetric_orthogonalization), "6D": (6, compute_rotation_matrix_from_ortho6d),
                    "3D": (3, vec_3d_to_SO3)}
head_dimensions = {"SVD": 9, "6D": 6, "5D": 5, "Quat": 4, "Euler": 3, "Direct": 9}
head_functions = {"SVD": symmetric_orthogonalization, "6D": compute_rotation_matrix_from_ortho6d,
                  "5D": compute_rotation_matrix_from_ortho5d, "Quat": compute_rotation_matrix_from_quaternion,
                  "quat": compute_rotation_matrix_from_quaternion, "Euler": compute_rotation_matrix_from_euler}
