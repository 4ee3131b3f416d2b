#!/usr/bin/env python3
"""Accuracy of the float32 projection on 1M Gaussian rows against numpy's float64 LAPACK SVD (no oracle: tools/ may not use it):
median / p99 / p99.9 / max of |dR| per row, the same scaled by the row's conditioning gap/s1, orthogonality, flip flags."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from poseestimation_amd import rotation_representation as rr

g = torch.Generator().manual_seed(0)
x = torch.randn(1_000_000, 9, generator=g)
r = rr.symmetric_orthogonalization(x.cuda()).cpu().numpy().astype(np.float64)
m = x.numpy().astype(np.float64).reshape(-1, 3, 3)
u, s, vt = np.linalg.svd(m)
d = np.sign(np.linalg.det(u @ vt))
u[:, :, 2] *= d[:, None]
ref = u @ vt
err = np.abs(r - ref).reshape(len(r), -1).max(1)
gap = np.where(d < 0, s[:, 1] - s[:, 2], s[:, 1] + s[:, 2]) / s[:, 0]
orth = np.abs(r @ r.transpose(0, 2, 1) - np.eye(3)).reshape(len(r), -1).max(1)
q = lambda a, p: float(np.quantile(a, p))
print("|dR| per row: median %.2e  p99 %.2e  p99.9 %.2e  max %.2e" % (np.median(err), q(err, 0.99), q(err, 0.999), err.max()))
print("|dR| gap/s1 : median %.2e  p99 %.2e  p99.9 %.2e  max %.2e" % (np.median(err * gap), q(err * gap, 0.99), q(err * gap, 0.999), (err * gap).max()))
print("|R R^T - I| max %.2e   det(R) min %.6f" % (orth.max(), np.linalg.det(r).min()))
