#!/usr/bin/env python3
"""frobenius_head at 1M rows through the mirror, with the reduction workspace (one launch, ticket finish, kernel-written mean) and without
(memset + kernel + mean kernel): device time per forward call by HIP events, rotating inputs.  One device, alternating."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poseestimation_amd import rotation_representation as rr

dev = "cuda:0"
n, NB = 1_000_000, 6
xs = [torch.randn(n, 9, device=dev) for _ in range(NB)]
ts = [rr.symmetric_orthogonalization(torch.randn(n, 9, device=dev)) for _ in range(NB)]
real_ws = rr._workspace


def run(want_r, grad, iters=60):
    for i in range(5):
        x = xs[i % NB].requires_grad_(grad)
        rr.frobenius_head(x, ts[i % NB], return_rotation=want_r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        x = xs[i % NB].requires_grad_(grad)
        rr.frobenius_head(x, ts[i % NB], return_rotation=want_r)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for rnd in range(3):
    for want_r in (True, False):
        for grad in (True, False):
            rr._workspace = real_ws
            a = min(run(want_r, grad) for _ in range(3))
            rr._workspace = lambda d, s: None
            b = min(run(want_r, grad) for _ in range(3))
            print("R %-5s dM %-5s  workspace %.2f us   no workspace (memset + kernel + mean) %.2f us" % (want_r, grad, a, b), flush=True)
rr._workspace = real_ws
