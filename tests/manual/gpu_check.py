#!/usr/bin/env python3
"""Quick GPU sanity + timing script used during development (not a test; tests/ holds the parity suite)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import poseestimation_amd as pa
from poseestimation_amd import rotation_representation as rr
from oracle import c_oracle

def main():
    dev = torch.device("cuda:0")
    print(torch.cuda.get_device_name(0))
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1_000_000, 9, generator=g)
    xd = x.to(dev)
    r, flip = rr.symmetric_orthogonalization_with_flip(xd)
    torch.cuda.synchronize()
    r_ref, f_ref = c_oracle.project(x.numpy(), want_flip=True)
    rc = r.cpu().numpy()
    err = np.abs(rc - r_ref).reshape(len(x), -1).max(1)
    orth = np.linalg.norm(np.einsum('bji,bjk->bik', rc, rc) - np.eye(3), axis=(1, 2))
    print("K1: max|dR| %.3e p99.9 %.3e median %.3e  orth max %.3e  flip mismatches %d  nan %d" % (
        err.max(), np.quantile(err, .999), np.median(err), orth.max(), (flip.cpu().numpy() != f_ref).sum(), np.isnan(rc).sum()))
    # angle error
    t = rr.symmetric_orthogonalization(torch.randn(1_000_000, 9, generator=torch.Generator().manual_seed(1)).to(dev))
    deg = rr.angle_error(r, t)
    sc = rr.angle_error_sum_count(r, t)
    deg_ref, _ = c_oracle.angle_error(rc, t.cpu().numpy())
    print("K4: mean %.9f (sum/count %.9f) oracle-on-same-R %.9f  max|d| %.3e" % (
        deg.mean().item(), (sc[0] / sc[1]).item(), deg_ref.mean(), np.abs(deg.cpu().numpy() - deg_ref).max()))
    deg_o, _ = c_oracle.angle_error(r_ref, c_oracle.project(torch.randn(1_000_000, 9, generator=torch.Generator().manual_seed(1)).numpy()))
    print("    mean angle, oracle end to end %.9f  delta %.3e deg" % (deg_o.mean(), deg.mean().item() - deg_o.mean()))
    # backward
    xg = xd[:4096].clone().requires_grad_(True)
    gg = torch.randn(4096, 3, 3, generator=torch.Generator().manual_seed(5)).to(dev)
    rr.symmetric_orthogonalization(xg).backward(gg)
    ref = c_oracle.project_bwd(x[:4096].numpy(), gg.cpu().numpy())
    e = np.abs(xg.grad.cpu().numpy().reshape(-1, 3, 3) - ref).reshape(4096, -1).max(1) / (1 + np.abs(ref).reshape(4096, -1).max(1))
    print("K2: rel err max %.3e p99 %.3e median %.3e" % (e.max(), np.quantile(e, .99), np.median(e)))
    # fused
    xf = xd[:512].clone().requires_grad_(True)
    loss, rf = rr.frobenius_head(xf, t[:512])
    loss.backward()
    xr = xd[:512].clone().requires_grad_(True)
    l2 = rr.loss_frobenius(t[:512], rr.symmetric_orthogonalization(xr)); l2.backward()
    print("K3: loss %.7f vs unfused %.7f ; grad max diff %.3e" % (loss.item(), l2.item(), (xf.grad - xr.grad).abs().max().item()))
    # kabsch
    B, N = 4096, 1024
    P = (torch.rand(B, N, 3, generator=g) - 0.5).to(dev)
    Rgt = rr.symmetric_orthogonalization(torch.randn(B, 9, generator=g).to(dev))
    Q = torch.bmm(P, Rgt.transpose(1, 2)) + 0.01 * torch.randn(B, N, 3, device=dev)
    Rk, H = rr.kabsch_rotation(P, Q, return_h=True)
    Ro, Ho = c_oracle.kabsch(P.cpu().numpy(), Q.cpu().numpy(), want_h=True)
    print("K5: max|dH| %.3e max|dR| %.3e  angle to gt mean %.4f deg" % (np.abs(H.cpu().numpy() - Ho).max(), np.abs(Rk.cpu().numpy() - Ro).max(), rr.angle_error(Rk, Rgt).mean().item()))

    # timing K1 with rotating buffers
    nb = 10
    xs = [torch.randn(1_000_000, 9, device=dev) for _ in range(nb)]
    outs = [torch.empty(1_000_000, 3, 3, device=dev) for _ in range(nb)]
    lib = pa._lib.load()
    import ctypes
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    def launch(i):
        lib.so3_project_fwd_f32(ctypes.c_void_p(xs[i % nb].data_ptr()), ctypes.c_void_p(outs[i % nb].data_ptr()), None, 1_000_000, st)
    for i in range(5): launch(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 50
    e0.record()
    for i in range(K): launch(i)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    print("K1 1M rotating: %.2f us/launch  %.2f Gproj/s  %.1f GB/s (%.1f%% of 8 TB/s)" % (ms * 1e3, 1e6 / ms / 1e6, 72e6 / ms / 1e6, 72e6 / ms / 1e6 / 8000 * 100))
    e0.record()
    for i in range(K): launch(0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    print("K1 1M same buffer (cache resident): %.2f us/launch  %.1f GB/s" % (ms * 1e3, 72e6 / ms / 1e6))
    # kabsch timing
    B = 65536
    P = torch.rand(B, 1024, 3, device=dev) - 0.5
    Q = torch.rand(B, 1024, 3, device=dev) - 0.5
    Rk = torch.empty(B, 3, 3, device=dev)
    def lk():
        lib.so3_kabsch_f32(ctypes.c_void_p(P.data_ptr()), ctypes.c_void_p(Q.data_ptr()), ctypes.c_void_p(Rk.data_ptr()), None, B, 1024, st)
    for i in range(3): lk()
    torch.cuda.synchronize()
    e0.record()
    for i in range(10): lk()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("K5 65536x1024: %.1f us  %.1f GB/s (%.1f%% of 8 TB/s)" % (ms * 1e3, B * 24612 / ms / 1e6, B * 24612 / ms / 1e6 / 8000 * 100))



def time_k4():
    import ctypes
    dev = torch.device("cuda:0")
    lib = pa._lib.load()
    n = 1_000_000
    a = [rr.symmetric_orthogonalization(torch.randn(n, 9, device=dev)) for _ in range(4)]
    b = [rr.symmetric_orthogonalization(torch.randn(n, 9, device=dev)) for _ in range(4)]
    deg = torch.empty(n, dtype=torch.float64, device=dev)
    sc = torch.empty(2, dtype=torch.float64, device=dev)
    fl = torch.empty(1, dtype=torch.int32, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, dptr, sptr in (("sum only", None, sc), ("deg only", deg, None), ("deg+sum", deg, sc)):
        def go(i):
            lib.so3_angle_error_v2(ctypes.c_void_p(a[i % 4].data_ptr()), ctypes.c_void_p(b[i % 4].data_ptr()), ctypes.c_void_p(dptr.data_ptr()) if dptr is not None else None, ctypes.c_void_p(sptr.data_ptr()) if sptr is not None else None, ctypes.c_void_p(fl.data_ptr()), None, 0, n, st)
        for i in range(3): go(i)
        torch.cuda.synchronize(); e0.record()
        for i in range(20): go(i)
        e1.record(); torch.cuda.synchronize()
        print("K4 1M %-9s %.2f us/call (incl. memset + count launches)" % (name, e0.elapsed_time(e1) / 20 * 1e3))




def time_config4():
    """Config #4: B = 512 bf16 head forward + Frobenius loss + backward, one fused call."""
    import ctypes
    dev = torch.device("cuda:0")
    lib = pa._lib.load()
    b = 512
    x = torch.randn(b, 9, device=dev).bfloat16()
    rt = rr.symmetric_orthogonalization(torch.randn(b, 9, device=dev))
    r = torch.empty(b, 9, device=dev); dm = torch.empty(b, 9, device=dev, dtype=torch.bfloat16)
    ls = torch.empty(1, dtype=torch.float64, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    def go():
        lib.so3_frob_fwd_bwd_v2_bf16(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(rt.data_ptr()), ctypes.c_void_p(r.data_ptr()), ctypes.c_void_p(dm.data_ptr()), ctypes.c_void_p(ls.data_ptr()), None, None, 0, b, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10): go()
    torch.cuda.synchronize(); e0.record()
    for _ in range(200): go()
    e1.record(); torch.cuda.synchronize()
    print("config4 fused C-ABI call (B=512, bf16): %.2f us/call back-to-back" % (e0.elapsed_time(e1) / 200 * 1e3))
    # through the Python mirror incl. autograd bookkeeping
    xg = x.clone().requires_grad_(True)
    import time
    for _ in range(5):
        loss, _ = rr.frobenius_head(xg, rt); loss.backward(); xg.grad = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        loss, _ = rr.frobenius_head(xg, rt); loss.backward(); xg.grad = None
    torch.cuda.synchronize()
    print("config4 via Python mirror fwd+bwd: %.1f us/iter (host-bound)" % ((time.perf_counter() - t0) / 100 * 1e6))
    # the reference's op chain on the same GPU, for scale (torch.linalg.svd on device)
    from oracle import so3_oracle as so
    xf = x.float().clone().requires_grad_(True)
    try:
        for _ in range(3):
            l = so.loss_frobenius_torch(so.symmetric_orthogonalization_torch(xf), rt); l.backward(); xf.grad = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            l = so.loss_frobenius_torch(so.symmetric_orthogonalization_torch(xf), rt); l.backward(); xf.grad = None
        torch.cuda.synchronize()
        print("config4 reference ATen chain on the same GPU (torch.linalg.svd): %.1f us/iter" % ((time.perf_counter() - t0) / 20 * 1e6))
    except Exception as exc:
        print("reference chain on GPU failed:", exc)


if __name__ == "__main__":
    main()
    time_k4()
    time_config4()
