#!/usr/bin/env python3
"""Launch K1 (1M rows, rotating buffers) N times -- the program rocprofv3 wraps for kernel traces / PMC passes.
usage: k1_loop.py [launches [rows [family share]]]     family: one of tools/k1_hard_rows.py's (e.g. "near-reflection", "generic ties"), mixed into the Gaussian rows"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poseestimation_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
lib = _lib.load()
dev = torch.device("cuda:0")
nb = 8
xs = [torch.randn(rows, 9, device=dev) for _ in range(nb)]
if len(sys.argv) > 4:
    import importlib.util
    spec = importlib.util.spec_from_file_location("k1_hard_rows", os.path.join(os.path.dirname(os.path.abspath(__file__)), "k1_hard_rows.py"))
    hr = importlib.util.module_from_spec(spec); spec.loader.exec_module(hr)
    gen = torch.Generator(device=dev).manual_seed(5)
    for x in xs:
        idx = torch.nonzero(torch.rand(rows, device=dev, generator=gen) < float(sys.argv[4])).flatten()
        x[idx] = hr.family(sys.argv[3], idx.numel(), dev, gen).reshape(-1, 9)
outs = [torch.empty(rows, 9, device=dev) for _ in range(nb)]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
for i in range(n):
    lib.so3_project_fwd_f32(ctypes.c_void_p(xs[i % nb].data_ptr()), ctypes.c_void_p(outs[i % nb].data_ptr()), None, rows, st)
torch.cuda.synchronize()
print("done", n)
