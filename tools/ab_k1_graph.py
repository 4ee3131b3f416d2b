#!/usr/bin/env python3
"""A/B of K1 between builds of the library: per build one hipGraph of 500 launches (1M rows, 8 rotating buffers);
builds are interleaved, each measured as the mean of 4 timed replays after 2 untimed ones.  usage: ab_k1_graph.py a.so b.so ..."""
import ctypes, sys
import torch

P = ctypes.c_void_p
dev = torch.device("cuda:0")
n, NB, K = 1_000_000, 8, 500
x = [torch.randn(n, 9, device=dev) for _ in range(NB)]
r = [torch.empty(n, 9, device=dev) for _ in range(NB)]
graphs = {}
side = torch.cuda.Stream()
for path in sys.argv[1:]:
    lib = ctypes.CDLL(path)
    lib.so3_project_fwd_f32.restype = ctypes.c_int
    lib.so3_project_fwd_f32.argtypes = [P, P, P, ctypes.c_int64, P]
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            st = P(side.cuda_stream)
            for i in range(K):
                assert lib.so3_project_fwd_f32(P(x[i % NB].data_ptr()), P(r[i % NB].data_ptr()), None, n, st) == 0
    graphs[path.split("/")[-1]] = (g, lib)
torch.cuda.synchronize()
names = list(graphs)
for rnd in range(int(__import__("os").environ.get("AB_ROUNDS", "4"))):
    line = []
    order = names[rnd % len(names):] + names[:rnd % len(names)]          # the order rotates: no build always runs behind the same one
    for name in order:
        g, _ = graphs[name]
        with torch.cuda.stream(side):
            g.replay(); g.replay()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(side)
            for _ in range(4):
                g.replay()
            b.record(side)
        torch.cuda.synchronize()
        line.append((name, a.elapsed_time(b) * 1e3 / (4 * K)))
    print("  ".join("%s %.2f" % (n_, dict(line)[n_]) for n_ in names), flush=True)
