#!/usr/bin/env python3
"""Are the streaming engine (packed, two rows per lane) and the one-row-per-lane tile kernels bit-identical?"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poseestimation_amd import rotation_representation as rr
dev = "cuda:0"
n = 200_000
g = torch.Generator(device=dev).manual_seed(5)
x = torch.randn(n, 9, device=dev, generator=g)
base = torch.empty(n * 9 + 1, device=dev)
xu = base[1:].view(n, 9); xu.copy_(x)
r_a = rr.symmetric_orthogonalization(x)
r_u = rr.symmetric_orthogonalization(xu)
print("K1 engine vs tile: max |diff| =", (r_a - r_u).abs().max().item(), " differing rows:", int(((r_a != r_u).flatten(1).any(1)).sum()))
t = rr.symmetric_orthogonalization(torch.randn(n, 9, device=dev, generator=g))
gup = torch.randn(n, 3, 3, device=dev, generator=g)
xa = x.clone().requires_grad_(True); rr.symmetric_orthogonalization(xa).backward(gup)
xb = xu.clone()  # aligned copy again; use the unaligned view for the leaf instead
leaf = torch.empty(n * 9 + 1, device=dev)[1:].view(n, 9); leaf.copy_(x); leaf.requires_grad_(True)
rr.symmetric_orthogonalization(leaf).backward(gup)
print("K2 engine vs tile: max |diff| =", (xa.grad - leaf.grad).abs().max().item(), " differing rows:", int(((xa.grad != leaf.grad).flatten(1).any(1)).sum()))
