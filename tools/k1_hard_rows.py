#!/usr/bin/env python3
"""Worst-case throughput of K1 (and of the fused training step K3): microseconds per 1M rows when 0 %, 1 %, 10 %, 100 %
of the rows are HARD for the quaternion fast path (they are redone by the Jacobi path inside the kernel).

Families: rank two (one singular value 0: not hard), near-reflections (-R + noise: all singular values close, det < 0),
small integers (entries in {-1, 0, 1}: ties and rank deficiency), 1e5 * Gaussian (outside the fast path's scale window:
prescaled, not hard since round 3), generic ties (s2 = s3, det < 0, general position), rank one, all zero (a dead head).  The reference (torch.svd -> LAPACK / gesvdj) has no such cliff: this table puts ours on record.

usage: k1_hard_rows.py [--lib path/to/libso3proj.so] [--rows N]      (prints a table; docs/history/profiles/r03_k1_hard_rows.txt)
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poseestimation_amd import _lib

P = ctypes.c_void_p


def load(path):
    if path is None:
        return _lib.load()
    lib = ctypes.CDLL(path)
    for name in ("so3_project_fwd_f32", "so3_frob_fwd_bwd_f32"):
        res, args = _lib.SYMBOLS[name]
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = args
    return lib


def haar(n, dev, gen):
    """Haar rotations from normalised Gaussian quaternions (elementwise: batched QR of a million 3x3 blocks takes rocSOLVER minutes)."""
    q = torch.randn(n, 4, device=dev, generator=gen)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=1).view(n, 3, 3)


def family(name, n, dev, gen):
    if name == "rank two":
        u, v = haar(n, dev, gen), haar(n, dev, gen)
        s = torch.rand(n, 3, device=dev, generator=gen) + 0.5
        s[:, 2] = 0.0
        return (u * s.unsqueeze(1)) @ v.transpose(1, 2)
    if name == "near-reflection":
        return -haar(n, dev, gen) + 1e-3 * torch.randn(n, 3, 3, device=dev, generator=gen)
    if name == "entries in {-1,0,1}":
        return torch.randint(-1, 2, (n, 3, 3), device=dev, generator=gen).float()
    if name == "1e5 * Gaussian":
        return 1e5 * torch.randn(n, 3, 3, device=dev, generator=gen)
    if name == "generic ties":           # U diag(s1, s2, -s2 (1 - 1e-6)) V^T: s2 = s3 with det < 0 in general position
        u, v = haar(n, dev, gen), haar(n, dev, gen)
        s = torch.rand(n, 3, device=dev, generator=gen) + 0.5
        s[:, 0] += 1.0
        s[:, 2] = -s[:, 1] * (1 - 1e-6)
        return (u * s.unsqueeze(1)) @ v.transpose(1, 2)
    if name == "all zero":                # a dead head: outside the scale window, every wave leaves the fast path after the quartic
        return torch.zeros(n, 3, 3, device=dev)
    if name == "rank one":
        a, b = torch.randn(n, 3, 1, device=dev, generator=gen), torch.randn(n, 1, 3, device=dev, generator=gen)
        return a @ b
    raise ValueError(name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--iters", type=int, default=200)
    args = ap.parse_args()
    lib = load(args.lib)
    dev = torch.device("cuda:0")
    n, nb = args.rows, 4
    gen = torch.Generator(device=dev).manual_seed(11)
    st = P(torch.cuda.current_stream().cuda_stream)
    out = [torch.empty(n, 9, device=dev) for _ in range(nb)]
    dm = [torch.empty(n, 9, device=dev) for _ in range(nb)]
    ls = torch.empty(1, dtype=torch.float64, device=dev)
    rt = haar(n, dev, gen).reshape(n, 9).contiguous()

    def timed(fn):
        for i in range(10):
            fn(i)
        torch.cuda.synchronize()
        best = float("inf")
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(args.iters):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / args.iters * 1e3)
        return best

    def bench(xs):
        k1 = timed(lambda i: lib.so3_project_fwd_f32(P(xs[i % nb].data_ptr()), P(out[i % nb].data_ptr()), None, n, st))
        k3 = timed(lambda i: lib.so3_frob_fwd_bwd_v2_f32(P(xs[i % nb].data_ptr()), P(rt.data_ptr()), P(out[i % nb].data_ptr()), P(dm[i % nb].data_ptr()), P(ls.data_ptr()), None, None, 0, n, st))
        return k1, k3

    gauss = [torch.randn(n, 9, device=dev, generator=gen) for _ in range(nb)]
    bench(gauss)                                   # the chip's clock ramps up over the first tens of milliseconds of load:
    g1, g3 = bench(gauss)                          # the reference batch is timed once the device is warm
    print("library: %s   rows: %d   (eager launches, %d rotating buffers, best of 3 x %d)" % (args.lib or _lib.LIB_PATH, n, nb, args.iters))
    print("%-22s %6s | %9s %7s | %9s %7s" % ("family of hard rows", "share", "K1 us", "x Gauss", "K3 us", "x Gauss"))
    print("%-22s %6s | %9.2f %7.2f | %9.2f %7.2f" % ("Gaussian (none hard)", "0 %", g1, 1.0, g3, 1.0))
    worst = 1.0
    for name in ("rank two", "near-reflection", "entries in {-1,0,1}", "1e5 * Gaussian", "generic ties", "rank one", "all zero"):
        for share in (0.01, 0.10, 1.0):
            xs = []
            for b in range(nb):
                x = gauss[b].clone()
                if share >= 1.0:
                    x = family(name, n, dev, gen).reshape(n, 9).contiguous()
                else:
                    idx = torch.nonzero(torch.rand(n, device=dev, generator=gen) < share).flatten()
                    x[idx] = family(name, idx.numel(), dev, gen).reshape(-1, 9)
                xs.append(x)
            k1, k3 = bench(xs)
            worst = max(worst, k1 / g1, k3 / g3)
            print("%-22s %5.0f %% | %9.2f %7.2f | %9.2f %7.2f" % (name, share * 100, k1, k1 / g1, k3, k3 / g3), flush=True)
            del xs
    print("worst ratio to the Gaussian batch: %.2f" % worst)


if __name__ == "__main__":
    main()
