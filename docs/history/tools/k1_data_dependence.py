#!/usr/bin/env python3
"""K1 on inputs of different bit activity, same instruction stream: Gaussian rows (the benchmark), one fixed well-conditioned
matrix repeated (every lane computes the same numbers: little switching), and the copy-rate reference (torch copy).  If the
kernel's time follows the data, the limiter is the clock the package grants (power), not instruction issue."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poseestimation_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
rows, nb = 1_000_000, 8
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = ctypes.c_void_p
outs = [torch.empty(rows, 9, device=dev) for _ in range(nb)]

def run(name, xs, reps=1500):
    for i in range(200):
        lib.so3_project_fwd_f32(P(xs[i % nb].data_ptr()), P(outs[i % nb].data_ptr()), None, rows, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        lib.so3_project_fwd_f32(P(xs[i % nb].data_ptr()), P(outs[i % nb].data_ptr()), None, rows, st)
    e1.record(); torch.cuda.synchronize()
    print("%-44s %.2f us per launch" % (name, e0.elapsed_time(e1) / reps * 1e3), flush=True)

g = [torch.randn(rows, 9, device=dev) for _ in range(nb)]
one = torch.tensor([[0.9, -0.2, 0.1, 0.3, 1.1, -0.4, -0.2, 0.5, 0.8]], device=dev)
c = [one.repeat(rows, 1).contiguous() for _ in range(nb)]
for rep in range(2):
    run("Gaussian rows", g)
    run("one matrix repeated 1M times", c)
    run("Gaussian rows", g)
