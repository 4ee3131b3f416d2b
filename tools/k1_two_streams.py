#!/usr/bin/env python3
"""K1 over independent batches on ONE stream against TWO: consecutive launches of the persistent engine leave the device half empty
while the waves that hold a third round finish (46 % of the waves of a 1M-row launch hold two) and while the next launch's first loads
are in flight; launches on a second stream fill those slots.  One hipGraph of K launches either way (8 rotating buffer pairs, 1M rows),
the two-stream graph forks after its first node and joins at the end.  us per launch = the graph's event time / K: with two streams that
is a THROUGHPUT figure (launches overlap; a profiler's per-kernel duration is longer).   usage: k1_two_streams.py [rows] [K]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poseestimation_amd import _lib

P = ctypes.c_void_p
dev = torch.device("cuda:0")
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 400
NB = 8
x = [torch.randn(n, 9, device=dev) for _ in range(NB)]
r = [torch.empty(n, 9, device=dev) for _ in range(NB)]


def launch(i, stream):
    assert lib.so3_project_fwd_f32(P(x[i % NB].data_ptr()), P(r[i % NB].data_ptr()), None, n, P(stream.cuda_stream)) == 0


def capture(nstreams):
    side = [torch.cuda.Stream() for _ in range(nstreams)]
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.stream(side[0]):
        with torch.cuda.graph(g, stream=side[0]):
            for s in side[1:]:
                s.wait_stream(side[0])                   # fork
            for i in range(K):
                launch(i, side[i % nstreams])
            for s in side[1:]:
                side[0].wait_stream(s)                   # join
    return g, side[0]


graphs = {k: capture(k) for k in (1, 2, 3)}
torch.cuda.synchronize()
for rnd in range(6):
    line = []
    for k, (g, s) in graphs.items():
        with torch.cuda.stream(s):
            g.replay(); g.replay()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s)
            for _ in range(4):
                g.replay()
            b.record(s)
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / (4 * K)
        line.append("%d stream%s %.2f us/launch (%.1f %% of 8 TB/s)" % (k, "s" if k > 1 else " ", us, 72.0 * n / us * 1e-3 / 80))
    print("   ".join(line), flush=True)
