#!/usr/bin/env python3
"""Is the cost of frobenius_head + backward through autograd stable within one process?  (It is bimodal: ~65 us or ~110 us
per step for stretches of a second -- where the autograd engine's device thread happens to be scheduled.)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from poseestimation_amd import rotation_representation as rr
dev = torch.device('cuda', 0)
b = 512
x4 = torch.randn(b, 9, device=dev).bfloat16()
t4 = rr.symmetric_orthogonalization(torch.randn(b, 9, device=dev))
xg = x4.clone().requires_grad_(True)
def mirror():
    loss, _r = rr.frobenius_head(xg, t4)
    loss.backward()
    xg.grad = None
for i in range(200): mirror()
out = []
for rep in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(1000): mirror()
    torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 1000 * 1e6)
print("1000-step blocks, us per step:", " ".join("%.0f" % v for v in out))
print("affinity:", len(os.sched_getaffinity(0)), "cpus; threads torch:", torch.get_num_threads())
if len(sys.argv) > 1:
    os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0], sorted(os.sched_getaffinity(0))[1]})
    out = []
    for rep in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(1000): mirror()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 1000 * 1e6)
    print("pinned to two cpus:", " ".join("%.0f" % v for v in out))
