#!/usr/bin/env python3
"""Launch one of the two-input kernels (1M rows, rotating buffers) N times -- the program rocprofv3 wraps for PMC passes.
usage: kernel_loop.py k14|k2|k3|k4 [launches] [rows]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poseestimation_amd import _lib
which = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
lib = _lib.load()
dev = torch.device("cuda:0")
nb = 8
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
xs = [torch.randn(rows, 9, device=dev) for _ in range(nb)]
gs = [torch.randn(rows, 9, device=dev) for _ in range(nb)]
rt = []
for x in xs:                                   # rotations for the second operand: the projections of other Gaussians
    r = torch.empty(rows, 9, device=dev)
    lib.so3_project_fwd_f32(p(torch.randn(rows, 9, device=dev)), p(r), None, rows, None)
    rt.append(r)
outs = [torch.empty(rows, 9, device=dev) for _ in range(nb)]
dms = [torch.empty(rows, 9, device=dev) for _ in range(nb)]
ls = torch.zeros(1, dtype=torch.float64, device=dev)
sc = torch.zeros(2, dtype=torch.float64, device=dev)
fl = torch.zeros(1, dtype=torch.int32, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
for i in range(n):
    j = i % nb
    if which == "k14":
        sc.zero_()
        lib.so3_project_angle_error_v2_f32(p(xs[j]), p(rt[j]), None, None, p(sc), p(fl), None, _lib.PREZEROED, rows, st)
    elif which == "k2":
        lib.so3_project_bwd_f32(p(xs[j]), p(gs[j]), p(dms[j]), rows, st)
    elif which == "k3":
        lib.so3_frob_fwd_bwd_v2_f32(p(xs[j]), p(rt[j]), p(outs[j]), p(dms[j]), p(ls), None, None, 0, rows, st)
    elif which == "k4":
        sc.zero_()
        lib.so3_angle_error_v2(p(outs[j] if False else rt[j]), p(rt[(j + 1) % nb]), None, p(sc), p(fl), None, _lib.PREZEROED, rows, st)
    else:
        raise SystemExit("which?")
torch.cuda.synchronize()
print("done", which, n)
