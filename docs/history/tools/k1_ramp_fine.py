#!/usr/bin/env python3
"""Clock ramp of K1 from idle at 0.3-ms resolution: a 20-launch hipGraph (1M rows, 8 rotating buffer pairs) replayed back to
back for ~0.4 s, one event pair per replay; prints us per launch averaged over 5-ms bins.  usage: k1_ramp_fine.py [idle_s]"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poseestimation_amd import _lib

idle = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
lib = _lib.load()
dev = torch.device("cuda:0")
rows, nb, per = 1_000_000, 8, 20
xs = [torch.randn(rows, 9, device=dev) for _ in range(nb)]
outs = [torch.empty(rows, 9, device=dev) for _ in range(nb)]
P = ctypes.c_void_p
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    st = P(s.cuda_stream)
    for i in range(5):
        lib.so3_project_fwd_f32(P(xs[i % nb].data_ptr()), P(outs[i % nb].data_ptr()), None, rows, st)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        for i in range(per):
            lib.so3_project_fwd_f32(P(xs[i % nb].data_ptr()), P(outs[i % nb].data_ptr()), None, rows, st)
    s.synchronize()
    g.replay(); s.synchronize()
    for trial in range(2):
        time.sleep(idle)
        n = 1300
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record(s)
        for r in range(n):
            g.replay(); ev[r + 1].record(s)
        s.synchronize()
        t = [ev[0].elapsed_time(e) for e in ev]                 # ms since start
        bins = {}
        for r in range(n):
            b = int(t[r + 1] // 5)
            bins.setdefault(b, []).append((t[r + 1] - t[r]) * 1e3 / per)
        print("trial %d after %.1f s idle; us/launch per 5-ms bin:" % (trial, idle))
        print(" ".join("%.2f" % (sum(v) / len(v)) for k, v in sorted(bins.items())))
