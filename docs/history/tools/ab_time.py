#!/usr/bin/env python3
"""A/B timing of the reduction kernels for alternative builds of the library.
usage: ab_time.py lib_a.so [lib_b.so ...]   -- each is dlopen'ed directly; 1M rows, 8 rotating buffers."""
import ctypes, sys
import torch

P = ctypes.c_void_p
dev = torch.device("cuda:0")
n, NB = 1_000_000, 8
x = [torch.randn(n, 9, device=dev) for _ in range(NB)]
rt = [torch.linalg.qr(torch.randn(n, 3, 3, device=dev))[0].reshape(n, 9).contiguous() for _ in range(NB)]
r = [torch.empty(n, 9, device=dev) for _ in range(NB)]
dm = [torch.empty(n, 9, device=dev) for _ in range(NB)]
ls = torch.empty(1, dtype=torch.float64, device=dev)
sc = torch.empty(2, dtype=torch.float64, device=dev)
fl = torch.empty(1, dtype=torch.int32, device=dev)
deg = torch.empty(n, dtype=torch.float64, device=dev)
st = P(torch.cuda.current_stream().cuda_stream)
p = lambda t: P(t.data_ptr())


def timeit(fn, iters=100, warm=10):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


for rep in range(3):
    for path in sys.argv[1:]:
        lib = ctypes.CDLL(path)
        lib.so3_project_fwd_f32.restype = ctypes.c_int
        lib.so3_project_fwd_f32.argtypes = [P, P, P, ctypes.c_int64, P]
        lib.so3_project_bwd_f32.restype = ctypes.c_int
        lib.so3_project_bwd_f32.argtypes = [P, P, P, ctypes.c_int64, P]
        for f in ("so3_angle_error", "so3_project_angle_error_f32", "so3_frob_loss_f32", "so3_frob_fwd_bwd_f32"):
            getattr(lib, f).restype = ctypes.c_int
        lib.so3_angle_error.argtypes = [P, P, P, P, P, ctypes.c_int, ctypes.c_int64, P]
        lib.so3_project_angle_error_f32.argtypes = [P, P, P, P, P, P, ctypes.c_int, ctypes.c_int64, P]
        lib.so3_frob_loss_f32.argtypes = [P, P, P, P, ctypes.c_int64, P]
        lib.so3_frob_fwd_bwd_f32.argtypes = [P, P, P, P, P, ctypes.c_int64, P]
        res = {
            "K1": timeit(lambda i: lib.so3_project_fwd_f32(p(x[i % NB]), p(r[i % NB]), None, n, st), iters=400, warm=100),
            "K2": timeit(lambda i: lib.so3_project_bwd_f32(p(x[i % NB]), p(rt[i % NB]), p(dm[i % NB]), n, st)),
            "K4 deg": timeit(lambda i: lib.so3_angle_error(p(r[i % NB]), p(rt[i % NB]), p(deg), None, p(fl), 0, n, st)),
            "K4 sum": timeit(lambda i: lib.so3_angle_error(p(rt[(i + 1) % NB]), p(rt[i % NB]), None, p(sc), p(fl), 0, n, st)),
            "K1+K4 sum": timeit(lambda i: lib.so3_project_angle_error_f32(p(x[i % NB]), p(rt[i % NB]), None, None, p(sc), p(fl), 0, n, st)),
            "K3' loss+grad": timeit(lambda i: lib.so3_frob_loss_f32(p(rt[(i + 1) % NB]), p(rt[i % NB]), p(dm[i % NB]), p(ls), n, st)),
            "K3 dM+R": timeit(lambda i: lib.so3_frob_fwd_bwd_f32(p(x[i % NB]), p(rt[i % NB]), p(r[i % NB]), p(dm[i % NB]), p(ls), n, st)),
        }
        print("%-40s " % path.split("/")[-1] + "  ".join("%s %.2f" % kv for kv in res.items()), flush=True)
