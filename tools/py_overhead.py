#!/usr/bin/env python3
"""Host-side price list of the primitives the Python mirror is made of (B = 512), in microseconds per call on this box:
what a small-batch step can and cannot shed."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poseestimation_amd import rotation_representation as rr

dev = torch.device("cuda:0")
x = torch.randn(512, 9, device=dev).bfloat16().requires_grad_(True)
xf = torch.randn(512, 9, device=dev)
t = torch.eye(3, device=dev).repeat(512, 1, 1)
r = torch.empty(512, 3, 3, device=dev)
g1 = torch.ones((), device=dev)


def us(fn, n=20000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return dt


class F1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b_):
        return a.new_empty(())

    @staticmethod
    def backward(ctx, g):
        return None, None


fwd = rr._fn("so3_frob_fwd_bwd_v2_bf16")
scale = rr._fn("so3_scale_bf16")
dm = torch.empty_like(x)
loss = torch.empty((), device=dev)
items = [
    ("torch.empty(())", lambda: torch.empty((), dtype=torch.float32, device=dev)),
    ("torch.empty((512,3,3))", lambda: torch.empty((512, 3, 3), dtype=torch.float32, device=dev)),
    ("torch.empty_like(x)", lambda: torch.empty_like(x)),
    ("x.is_cuda + device compare", lambda: x.is_cuda and t.is_cuda and x.device == t.device),
    ("_head_input(x)", lambda: rr._head_input(x)),
    ("t.is_contiguous() + numel + dtype", lambda: t.dtype is torch.float32 and t.is_contiguous() and t.numel() == 4608),
    ("_stream(dev)", lambda: rr._stream(dev)),
    ("_on_device(dev) with-block", lambda: rr._on_device(dev).__enter__()),
    ("data_ptr() x4", lambda: (x.data_ptr(), t.data_ptr(), r.data_ptr(), dm.data_ptr())),
    ("K3 launch through _so3fast (enqueue only)", lambda: fwd(x.data_ptr(), t.data_ptr(), r.data_ptr(), dm.data_ptr(), None, loss.data_ptr(), None, 0, 512, rr._stream(dev))),
    ("scale launch through _so3fast", lambda: scale(dm.data_ptr(), g1.data_ptr(), dm.data_ptr(), 4608, rr._stream(dev))),
    ("torch mul launch (dm * g)", lambda: dm * g1),
    ("Function.apply, 2 inputs, 1 output, no grad path", lambda: F1.apply(xf, t)),
    ("Function.apply, input requires grad", lambda: F1.apply(x, t)),
    ("frobenius_head forward only (requires grad)", lambda: rr.frobenius_head(x, t)),
    ("frobenius_head under no_grad-like (x.detach())", lambda: rr.frobenius_head(x.detach(), t)),
    ("symmetric_orthogonalization(xf) no grad", lambda: rr.symmetric_orthogonalization(xf)),
]
for name, fn in items:
    print("%-52s %7.2f us" % (name, us(fn, 5000)), flush=True)
