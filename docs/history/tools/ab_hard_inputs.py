#!/usr/bin/env python3
"""K1 on all-hard inputs, builds interleaved on one device: usage ab_hard_inputs.py a.so b.so ...  (us per 1M rows, best of 5 x 100 eager launches)"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poseestimation_amd import _lib
from k1_hard_rows import family

P = ctypes.c_void_p
dev = torch.device("cuda:0")
n, NB = 1_000_000, 6
gen = torch.Generator(device=dev).manual_seed(3)
st = P(torch.cuda.current_stream().cuda_stream)
libs = {}
for path in sys.argv[1:]:
    lib = ctypes.CDLL(path)
    res, args = _lib.SYMBOLS["so3_project_fwd_f32"]
    lib.so3_project_fwd_f32.restype, lib.so3_project_fwd_f32.argtypes = res, args
    libs[os.path.basename(path).replace("libso3proj_", "").replace(".so", "")] = lib
out = [torch.empty(n, 9, device=dev) for _ in range(NB)]
inputs = {"gaussian": lambda: torch.randn(n, 9, device=dev, generator=gen), "zeros": lambda: torch.zeros(n, 9, device=dev)}
for name in ("near-reflection", "rank one", "generic ties", "entries in {-1,0,1}"):
    inputs[name] = (lambda nm: (lambda: family(nm, n, dev, gen).reshape(n, 9).contiguous()))(name)


def timed(fn, iters=100):
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


for name, make in inputs.items():
    xs = [make() for _ in range(NB)]
    best = {k: 1e9 for k in libs}
    for _ in range(5):
        for k, lib in libs.items():
            best[k] = min(best[k], timed(lambda i: lib.so3_project_fwd_f32(P(xs[i % NB].data_ptr()), P(out[i % NB].data_ptr()), None, n, st)))
    print("%-22s " % name + "   ".join("%s %.2f" % kv for kv in best.items()), flush=True)
    del xs
