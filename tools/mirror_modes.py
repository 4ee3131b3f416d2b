#!/usr/bin/env python3
"""Where the time of a launch-bound autograd step goes (config #4, B = 512 bf16): per-step host times of
  floor      an autograd.Function that launches nothing,
  floor+c    the same with one ctypes call (so3_version) in forward and backward through CDLL (releases the GIL),
  floor+p    the same through PyDLL (keeps the GIL),
  mirror     frobenius_head(x, t)[0].backward()
  two calls  loss_frobenius(t, symmetric_orthogonalization(x)).backward(), the reference's own spelling
as median / p10 / p90 of 20 blocks of 200 steps each (the cost is bimodal with where the autograd engine's device thread runs)."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poseestimation_amd import _lib, rotation_representation as rr

dev = "cuda:0"
lib = _lib.load()
plib = ctypes.PyDLL(_lib.LIB_PATH)
plib.so3_version.restype = ctypes.c_int


def make(call):
    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, buf):
            ctx.buf = buf
            if call is not None:
                call()
            return x.new_empty(())

        @staticmethod
        def backward(ctx, g):
            if call is not None:
                call()
            return ctx.buf, None
    return F


x = torch.randn(512, 9, device=dev).bfloat16().requires_grad_(True)
buf = torch.zeros_like(x)
t = torch.eye(3, device=dev).repeat(512, 1, 1)


def stepper(F):
    def f():
        F.apply(x, buf).backward()
        x.grad = None
    return f


def mirror():
    rr.frobenius_head(x, t)[0].backward()
    x.grad = None


def two_calls():                                   # the reference's own spelling (3D-Pose/main.py:60,85,90)
    rr.loss_frobenius(t, rr.symmetric_orthogonalization(x)).backward()
    x.grad = None


def blocks(fn, nblocks=20, n=200):
    for _ in range(300):
        fn()
    out = []
    for _ in range(nblocks):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / n * 1e6)
    out.sort()
    return out


for name, fn in (("floor", stepper(make(None))), ("floor + CDLL call", stepper(make(lib.so3_version))),
                 ("floor + PyDLL call", stepper(make(plib.so3_version))), ("mirror", mirror), ("floor (again)", stepper(make(None))),
                 ("mirror (again)", mirror), ("two calls (head, loss)", two_calls)):
    b = blocks(fn)
    print("%-22s median %6.1f  p10 %6.1f  p90 %6.1f  min %6.1f us/step" % (name, 0.5 * (b[9] + b[10]), b[2], b[17], b[0]), flush=True)
node, rr._so3node = rr._so3node, None              # the same spellings through the Python autograd.Functions
for name, fn in (("mirror, Python class", mirror), ("two calls, Python classes", two_calls)):
    b = blocks(fn)
    print("%-26s median %6.1f  p10 %6.1f  p90 %6.1f  min %6.1f us/step" % (name, 0.5 * (b[9] + b[10]), b[2], b[17], b[0]), flush=True)
rr._so3node = node
if len(sys.argv) > 1:
    os.sched_setaffinity(0, {0, 1})
    print("-- pinned to cores 0-1")
    for name, fn in (("floor", stepper(make(None))), ("mirror", mirror)):
        b = blocks(fn)
        print("%-22s median %6.1f  p10 %6.1f  p90 %6.1f  min %6.1f us/step" % (name, 0.5 * (b[9] + b[10]), b[2], b[17], b[0]), flush=True)
