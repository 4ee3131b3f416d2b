#!/usr/bin/env python3
"""Probe: symmetric input, float32 backward vs float64 backward -- find the worst rows."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poseestimation_amd import rotation_representation as rr
from oracle import so3_oracle as so
dev = "cuda:0"
torch.set_printoptions(precision=9, linewidth=220)
n = 2_000_000
worst_overall = 0
for seed in range(1, 9):
    g = torch.Generator(device=dev).manual_seed(1000 + seed)
    a = torch.randn(n, 3, 3, device=dev, generator=g, dtype=torch.float64)
    m32 = (a + a.transpose(1, 2)).float()
    gup = torch.randn(n, 3, 3, device=dev, generator=g)
    x32 = m32.clone().requires_grad_(True); rr.symmetric_orthogonalization(x32).backward(gup)
    x64 = m32.double().requires_grad_(True); rr.symmetric_orthogonalization(x64).backward(gup.double())
    s = torch.linalg.svdvals(m32.double())
    det = torch.linalg.det(m32.double())
    gap = torch.where(det < 0, s[:, 1] - s[:, 2], s[:, 1] + s[:, 2]) / s[:, 0]
    err = (x32.grad.double() - x64.grad).abs().flatten(1).amax(1) * s[:, 0] * gap * gap
    i = int(err.argmax())
    print("seed", seed, "worst", err[i].item(), "row", i, "s", s[i].tolist(), "det", det[i].item(), "gap", gap[i].item())
    if err[i].item() > 1e-3:
        mi = m32[i].cpu().numpy().astype(np.float64); gi = gup[i].cpu().numpy().astype(np.float64)
        print(" M =", m32[i].flatten().tolist())
        print(" G =", gup[i].flatten().tolist())
        print(" dM f32 =", x32.grad[i].flatten().tolist())
        print(" dM f64 =", x64.grad[i].flatten().tolist())
        print(" dM closed form (numpy f64) =", so.projection_backward_np(mi.reshape(1, 9), gi.reshape(1, 9)).ravel().tolist())
        r32 = rr.symmetric_orthogonalization(m32[i:i+1]); r64 = rr.symmetric_orthogonalization(m32[i:i+1].double())
        print(" R f32 =", r32.flatten().tolist()); print(" R f64 =", r64.flatten().tolist())
