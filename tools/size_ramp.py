#!/usr/bin/env python3
"""Time per launch against the batch size for the projection kernels of one build: the slope is the streaming rate, the intercept what a
launch costs besides streaming (launch boundary, filling the pipeline, the last round's arithmetic, the reduction's epilogue).
usage: size_ramp.py [lib.so]        prints one table (us per launch, best of 3 x 60 eager launches over rotating buffers)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _abi import Lib, P

dev = torch.device("cuda:0")
L = Lib(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "poseestimation_amd", "libso3proj.so"))
st = P(torch.cuda.current_stream().cuda_stream)
p = lambda t: P(t.data_ptr())
ws = torch.zeros(32768, dtype=torch.uint8, device=dev)
pool = torch.zeros(4096, 4, dtype=torch.float64, device=dev)
slot = lambda i: (P(pool[i % 4096].data_ptr()), P(pool[i % 4096].data_ptr() + 16))
ls = torch.empty(1, dtype=torch.float64, device=dev)
lm = torch.empty((), device=dev)


def timed(fn, iters=60):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


sizes = [256_000, 512_000, 1_000_000, 2_000_000, 4_000_000, 8_000_000]
rows = {}
for n in sizes:
    NB = max(3, min(8, int(600e6 / (n * 36)) + 1))
    x = [torch.randn(n, 9, device=dev) for _ in range(NB)]
    r = [torch.empty(n, 9, device=dev) for _ in range(NB)]
    t = [torch.empty(n, 9, device=dev) for _ in range(NB)]
    for a, b in zip(x, t):
        L.k1(p(a), p(b), n, st)                      # rotations as second inputs
    kern = {
        "K1 (72 B/row)": lambda i: L.k1(p(x[i % NB]), p(r[i % NB]), n, st),
        "K4 sum (72 B read)": lambda i: L.k4_sum(p(t[i % NB]), p(t[(i + 1) % NB]), slot(i)[0], slot(i)[1], n, st),
        "K1+K4 sum (72 B read)": lambda i: L.k14_sum(p(x[i % NB]), p(t[i % NB]), slot(i)[0], slot(i)[1], n, st),
        "K3 dM (108 B)": lambda i: L.k3(p(x[i % NB]), p(t[i % NB]), None, p(r[i % NB]), p(ls), p(lm), p(ws), n, st),
        "K2 (108 B)": lambda i: L.k2(p(x[i % NB]), p(t[i % NB]), p(r[i % NB]), n, st),
    }
    for k, fn in kern.items():
        rows.setdefault(k, []).append(min(timed(fn) for _ in range(3)))
    del x, r, t
    torch.cuda.empty_cache()
print("%-24s" % "rows" + "".join("%10d" % n for n in sizes) + "   slope us/M  intercept us   streaming TB/s")
for k, v in rows.items():
    b = float(k.split("(")[1].split()[0])
    A = np.vstack([np.array(sizes) / 1e6, np.ones(len(sizes))]).T
    slope, icpt = np.linalg.lstsq(A[2:], np.array(v)[2:], rcond=None)[0]          # fitted on 1M rows and up
    print("%-24s" % k + "".join("%10.2f" % t_ for t_ in v) + "   %9.2f  %12.2f   %8.2f" % (slope, icpt, b / slope), flush=True)
