#!/usr/bin/env python3
"""A/B of K1 / K2 / K3 between builds of the library on ONE device (devices differ by 10 %): builds are interleaved, each kernel
timed as the best of AB_ROUNDS x 100 eager launches over rotating buffers.  usage: ab_kernels.py a.so b.so ..."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poseestimation_amd import _lib

P = ctypes.c_void_p
dev = torch.device("cuda:0")
n, NB = 1_000_000, 6
x = [torch.randn(n, 9, device=dev) for _ in range(NB)]
g = [torch.randn(n, 9, device=dev) for _ in range(NB)]
r = [torch.empty(n, 9, device=dev) for _ in range(NB)]
dm = [torch.empty(n, 9, device=dev) for _ in range(NB)]
ls = torch.empty(1, dtype=torch.float64, device=dev)
lm = torch.empty((), device=dev)
deg = torch.empty(n, dtype=torch.float64, device=dev)
st = P(torch.cuda.current_stream().cuda_stream)
p = lambda t: P(t.data_ptr())
libs = {}
for path in sys.argv[1:]:
    lib = ctypes.CDLL(path)
    for name in ("so3_project_fwd_f32", "so3_project_bwd_f32", "so3_frob_fwd_bwd_ws_f32", "so3_reduce_workspace_bytes", "so3_project_angle_error_acc_f32",
                 "so3_angle_error_acc", "so3_angle_error", "so3_frob_loss_ws_f32", "so3_frob_loss_f32", "so3_frob_fwd_bwd_f32"):
        res, args = _lib.SYMBOLS[name]
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = args
    libs[os.path.basename(path).replace("libso3proj_", "").replace(".so", "")] = lib
ws = torch.zeros(32768, dtype=torch.uint8, device=dev)
pool = torch.zeros(4096, 4, dtype=torch.float64, device=dev)
rt = r[0].clone()
list(libs.values())[0].so3_project_fwd_f32(p(torch.randn(n, 9, device=dev)), p(rt), None, n, st)
rots = [rt, rt.clone()]
big = torch.zeros(4096, dtype=torch.float64, device=dev)
list(libs.values())[0].so3_project_fwd_f32(p(torch.randn(n, 9, device=dev)), p(rots[1]), None, n, st)


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


kernels = {
    "K1": lambda lib: (lambda i: lib.so3_project_fwd_f32(p(x[i % NB]), p(r[i % NB]), None, n, st)),
    "K2": lambda lib: (lambda i: lib.so3_project_bwd_f32(p(x[i % NB]), p(g[i % NB]), p(dm[i % NB]), n, st)),
    "K3 R+dM": lambda lib: (lambda i: lib.so3_frob_fwd_bwd_ws_f32(p(x[i % NB]), p(rt), p(r[i % NB]), p(dm[i % NB]), p(ls), p(lm), p(ws), n, st)),
    "K1+K4": lambda lib: (lambda i: lib.so3_project_angle_error_acc_f32(p(x[i % NB]), p(rt), None, None, P(pool[i % 4096].data_ptr()), P(pool[i % 4096].data_ptr() + 16), 0, n, st)),
    "K4 acc": lambda lib: (lambda i: lib.so3_angle_error_acc(p(rots[i % 2]), p(rots[1 - i % 2]), None, P(big.data_ptr()), None, 0, n, st)),
    "K4 deg": lambda lib: (lambda i: lib.so3_angle_error(p(rots[i % 2]), p(rots[1 - i % 2]), p(deg), None, None, 0, n, st)),
    "K3' ws": lambda lib: (lambda i: lib.so3_frob_loss_ws_f32(p(x[i % NB]), p(g[i % NB]), p(dm[i % NB]), p(ls), p(lm), p(ws), n, st)),
    "K3' at": lambda lib: (lambda i: lib.so3_frob_loss_f32(p(x[i % NB]), p(g[i % NB]), p(dm[i % NB]), p(ls), n, st)),
    "K3 at": lambda lib: (lambda i: lib.so3_frob_fwd_bwd_f32(p(x[i % NB]), p(rt), p(r[i % NB]), p(dm[i % NB]), p(ls), n, st)),
    "K3 dM": lambda lib: (lambda i: lib.so3_frob_fwd_bwd_ws_f32(p(x[i % NB]), p(rt), None, p(dm[i % NB]), p(ls), p(lm), p(ws), n, st)),
}
best = {(k, name): 1e9 for k in kernels for name in libs}
for rnd in range(int(os.environ.get("AB_ROUNDS", "4"))):
    for k, mk in kernels.items():
        if os.environ.get("AB_ONLY") and k not in os.environ["AB_ONLY"].split(","): continue
        for name, lib in libs.items():
            best[(k, name)] = min(best[(k, name)], timed(mk(lib)))
for k in kernels:
    if best[(k, list(libs)[0])] > 1e8: continue
    print("%-8s " % k + "   ".join("%s %.2f" % (name, best[(k, name)]) for name in libs), flush=True)
