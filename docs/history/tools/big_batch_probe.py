#!/usr/bin/env python3
"""120M-row probe (arrays of 4.32 GB: byte offsets cross 2^32), one progress line per step."""
import faulthandler, os, sys
import numpy as np
import torch
faulthandler.enable()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poseestimation_amd import rotation_representation as rr

def say(*a):
    print(*a, flush=True)

DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120_000_001
gen = torch.Generator(device=DEV).manual_seed(9)
x = torch.empty(n, 9, device=DEV)
for lo in range(0, n, 20_000_000):
    hi = min(n, lo + 20_000_000)
    x[lo:hi] = torch.randn(hi - lo, 9, device=DEV, generator=gen)
torch.cuda.synchronize(); say("filled", n)
r, flip = rr.symmetric_orthogonalization_with_flip(x)
torch.cuda.synchronize(); say("projected")
worst = 0.0
eye = torch.eye(3, device=DEV)
for lo in range(0, n, 20_000_000):
    blk = r[lo:lo + 20_000_000]
    e = torch.zeros(blk.shape[0], device=DEV)
    for i in range(3):
        for j in range(3):
            d = (blk[:, :, i] * blk[:, :, j]).sum(1) - (1.0 if i == j else 0.0)
            e += d * d
    worst = max(worst, e.sqrt().max().item())
    say("orth slab", lo, worst)
det_neg_ok = True
for lo in range(0, n, 20_000_000):
    m = x[lo:lo + 20_000_000].view(-1, 3, 3).double()
    det = (m[:, 0, 0] * (m[:, 1, 1] * m[:, 2, 2] - m[:, 1, 2] * m[:, 2, 1]) - m[:, 0, 1] * (m[:, 1, 0] * m[:, 2, 2] - m[:, 1, 2] * m[:, 2, 0])
           + m[:, 0, 2] * (m[:, 1, 0] * m[:, 2, 1] - m[:, 1, 1] * m[:, 2, 0]))
    bad = int(((det < 0) != flip[lo:lo + 20_000_000]).sum())
    say("flip slab", lo, "mismatches", bad)
sc = rr.angle_error_sum_count(r, r)
say("angle sum/count", sc[0].item(), sc[1].item())
say("DONE worst orth", worst)
