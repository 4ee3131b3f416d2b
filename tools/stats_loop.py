#!/usr/bin/env python3
"""Calls so3_angle_stats N times on 1M angles in 10 classes -- the program rocprofv3 wraps for its kernel trace.
usage: stats_loop.py [reps [classes [uniform|haar]]]     haar: K4's angles between two batches of random rotations (what tools/bench_all.py times)"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poseestimation_amd import rotation_representation as rr
n = 1_000_000
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ncls = int(sys.argv[2]) if len(sys.argv) > 2 else 10
kind = sys.argv[3] if len(sys.argv) > 3 else "uniform"
if kind == "haar":
    a, b = (rr.symmetric_orthogonalization(torch.randn(n, 9, device="cuda")) for _ in range(2))
    deg = rr.angle_error(a, b).to(torch.float64)
else:
    deg = torch.rand(n, device="cuda", dtype=torch.float64) * 180
cls = torch.randint(0, ncls, (n,), device="cuda", dtype=torch.int32)
for _ in range(3):
    rr.angle_error_statistics(deg, cls, ncls)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    rr.angle_error_statistics(deg, cls, ncls)
e1.record()
torch.cuda.synchronize()
print("so3_angle_stats, 1M %s angles, %d classes: %.1f us per call" % (kind, ncls, e0.elapsed_time(e1) / reps * 1e3))
