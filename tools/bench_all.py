#!/usr/bin/env python3
"""Times every C-ABI entry point at the BASELINE.json configs (secondary kernels; bench.py is the headline).
Each call is launched back-to-back on rotating buffers; reported per call with the algorithmic bytes it moves."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poseestimation_amd import _lib
from poseestimation_amd import rotation_representation as rr

P = ctypes.c_void_p
dev = torch.device("cuda:0")
if os.environ.get("SO3_LIB"):                    # another build of the library (tools/build_variant.sh), for A/B runs
    _lib.LIB_PATH = os.path.abspath(os.environ["SO3_LIB"])
lib = _lib.load()
ONLY = os.environ.get("SO3_BENCH_ONLY")          # time only the lines whose name contains this
st = P(torch.cuda.current_stream().cuda_stream)
NB = 6


QUICK = os.environ.get("SO3_BENCH_QUICK") == "1"      # 2 calls per entry: for rocprofv3 --pmc passes (kernels are serialised)


def timeit(name, fn, bytes_per_call, iters=60, warm=5):
    if ONLY and ONLY not in name:
        return
    if QUICK:
        iters, warm = 2, 1
    us = float("inf")
    for rep in range(1 if QUICK else 3):          # the best of three blocks: a line's first block reads up to 2 us high after a line of another kernel
        for i in range(warm): fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters): fn(i)
        e1.record(); torch.cuda.synchronize()
        us = min(us, e0.elapsed_time(e1) / iters * 1e3)
    print("%-58s %9.2f us/call  %7.0f GB/s (%4.1f%% of 8 TB/s)" % (name, us, bytes_per_call / us * 1e-3, bytes_per_call / us * 1e-3 / 80))


def main():
    n = 1_000_000
    x = [torch.randn(n, 9, device=dev) for _ in range(NB)]
    xb = [t.bfloat16() for t in x]
    g = [torch.randn(n, 9, device=dev) for _ in range(NB)]
    r = [torch.empty(n, 9, device=dev) for _ in range(NB)]
    dm = [torch.empty(n, 9, device=dev) for _ in range(NB)]
    dmb = [torch.empty(n, 9, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
    rt = [rr.symmetric_orthogonalization(torch.randn(n, 9, device=dev)).reshape(n, 9) for _ in range(NB)]
    flip = torch.empty(n, dtype=torch.uint8, device=dev)
    ls = torch.empty(1, dtype=torch.float64, device=dev)
    deg = torch.empty(n, dtype=torch.float64, device=dev)
    th = torch.empty(n, dtype=torch.float32, device=dev)
    sc = torch.empty(2, dtype=torch.float64, device=dev)
    fl = torch.empty(1, dtype=torch.int32, device=dev)
    p = lambda t: P(t.data_ptr())
    # the chip's clock needs tens of milliseconds of load to come up: keep it busy for ~80 ms before the first line
    import time
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.08:
        for i in range(64):
            lib.so3_project_fwd_f32(p(x[i % NB]), p(r[i % NB]), None, n, st)
        torch.cuda.synchronize()
    print("--- 1M rows (config #2 shape) ---")
    timeit("K1 so3_project_fwd_f32", lambda i: lib.so3_project_fwd_f32(p(x[i % NB]), p(r[i % NB]), None, n, st), 72 * n)
    timeit("K1 so3_project_fwd_f32 + flip flags", lambda i: lib.so3_project_fwd_f32(p(x[i % NB]), p(r[i % NB]), p(flip), n, st), 73 * n)
    timeit("K1 so3_project_fwd_bf16 (bf16 in, f32 out)", lambda i: lib.so3_project_fwd_bf16(p(xb[i % NB]), p(r[i % NB]), None, n, st), 54 * n)
    timeit("K2 so3_project_bwd_f32", lambda i: lib.so3_project_bwd_f32(p(x[i % NB]), p(g[i % NB]), p(dm[i % NB]), n, st), 108 * n)
    timeit("K2 so3_project_bwd_bf16", lambda i: lib.so3_project_bwd_bf16(p(xb[i % NB]), p(g[i % NB]), p(dmb[i % NB]), n, st), 72 * n)
    # the reducing entry points exist once (round 4): workspace nullable, a flags word.  "atomics" = no workspace (a zero-fill launch in front,
    # one atomic per workgroup behind); "workspace" = caller-owned, one launch, what the Python mirror passes for K3 / K3';
    # "zeroed slots" = SO3_PREZEROED, one launch, what the mirror passes for the metrics
    ws = torch.zeros(lib.so3_reduce_workspace_bytes(), dtype=torch.uint8, device=dev)
    lm = torch.empty((), dtype=torch.float32, device=dev)
    pool = torch.zeros(4096, 4, dtype=torch.float64, device=dev)
    slot = lambda i: (P(pool[i % 4096].data_ptr()), P(pool[i % 4096].data_ptr() + 16))
    PZ, EX = _lib.PREZEROED, _lib.EXACT_F64
    k3, k3p, k4, k14 = lib.so3_frob_fwd_bwd_v2_f32, lib.so3_frob_loss_v2_f32, lib.so3_angle_error_v2, lib.so3_project_angle_error_v2_f32
    timeit("K3 so3_frob_fwd_bwd_v2_f32 (R + dM + loss, atomics)", lambda i: k3(p(x[i % NB]), p(rt[i % NB]), p(r[i % NB]), p(dm[i % NB]), p(ls), None, None, 0, n, st), 144 * n)
    timeit("K3 so3_frob_fwd_bwd_v2_f32 (dM + loss, atomics)", lambda i: k3(p(x[i % NB]), p(rt[i % NB]), None, p(dm[i % NB]), p(ls), None, None, 0, n, st), 108 * n)
    timeit("K3 so3_frob_fwd_bwd_v2_f32 (R + dM + loss + mean, workspace)", lambda i: k3(p(x[i % NB]), p(rt[i % NB]), p(r[i % NB]), p(dm[i % NB]), p(ls), p(lm), p(ws), 0, n, st), 144 * n)
    timeit("K3 so3_frob_fwd_bwd_v2_f32 (dM + loss + mean, workspace)", lambda i: k3(p(x[i % NB]), p(rt[i % NB]), None, p(dm[i % NB]), p(ls), p(lm), p(ws), 0, n, st), 108 * n)
    timeit("K3' so3_frob_loss_v2_f32 (loss + dRpred, atomics)", lambda i: k3p(p(r[i % NB]), p(rt[i % NB]), p(dm[i % NB]), p(ls), None, None, 0, n, st), 108 * n)
    timeit("K3' so3_frob_loss_v2_f32 (loss + dRpred + mean, workspace)", lambda i: k3p(p(r[i % NB]), p(rt[i % NB]), p(dm[i % NB]), p(ls), p(lm), p(ws), 0, n, st), 108 * n)
    timeit("K4 so3_angle_error_v2 (per-row deg)", lambda i: k4(p(r[i % NB]), p(rt[i % NB]), p(deg), None, p(fl), None, 0, n, st), 80 * n)
    timeit("K4 so3_angle_error_v2 (sum,count, zeroed slots)", lambda i: k4(p(r[i % NB]), p(rt[i % NB]), None, slot(i)[0], slot(i)[1], None, PZ, n, st), 72 * n)
    timeit("K4 so3_angle_error_v2 (sum,count, init launch + atomics)", lambda i: k4(p(r[i % NB]), p(rt[i % NB]), None, p(sc), p(fl), None, 0, n, st), 72 * n)
    timeit("K1+K4 so3_project_angle_error_v2_f32 (sum: f32 outside the band, zeroed slots)", lambda i: k14(p(x[i % NB]), p(rt[i % NB]), None, None, slot(i)[0], slot(i)[1], None, PZ, n, st), 72 * n)
    timeit("K1+K4 so3_project_angle_error_v2_f32 (sum: float64 on every row, zeroed slots)", lambda i: k14(p(x[i % NB]), p(rt[i % NB]), None, None, slot(i)[0], slot(i)[1], None, PZ | EX, n, st), 72 * n)
    timeit("K1+K4 so3_project_angle_error_v2_f32 (sum: f32 outside the band, workspace)", lambda i: k14(p(x[i % NB]), p(rt[i % NB]), None, None, p(sc), p(fl), p(ws), 0, n, st), 72 * n)
    timeit("K1+K4 so3_project_angle_error_v2_f32 (per-row deg: float64)", lambda i: k14(p(x[i % NB]), p(rt[i % NB]), None, p(deg), None, p(fl), None, 0, n, st), 80 * n)
    # float64 arguments (the reference's metric casts to double itself; callers that already hold double rotations): one launch with the workspace
    r64 = [t.double() for t in rt[:2]]
    g64 = torch.empty(n, 9, dtype=torch.float64, device=dev)
    lm64 = torch.empty((), dtype=torch.float64, device=dev)
    timeit("K4 f64 so3_angle_error_v2_f64 (sum,count, workspace: one launch)", lambda i: lib.so3_angle_error_v2_f64(p(r64[i % 2]), p(r64[1 - i % 2]), None, p(sc), p(fl), p(ws), 0, n, st), 144 * n)
    timeit("K4 f64 so3_angle_error_v2_f64 (sum,count, no workspace: init + kernel)", lambda i: lib.so3_angle_error_v2_f64(p(r64[i % 2]), p(r64[1 - i % 2]), None, p(sc), p(fl), None, 0, n, st), 144 * n)
    timeit("K4 f64 so3_angle_error_v2_f64 (per-row deg)", lambda i: lib.so3_angle_error_v2_f64(p(r64[i % 2]), p(r64[1 - i % 2]), p(deg), None, p(fl), p(ws), 0, n, st), 152 * n)
    timeit("K3' f64 so3_frob_loss_v2_f64 (loss + dRpred, workspace: one launch)", lambda i: lib.so3_frob_loss_v2_f64(p(r64[i % 2]), p(r64[1 - i % 2]), p(g64), p(ls), p(lm64), p(ws), 0, n, st), 216 * n)
    timeit("K4' geodesic(R1, R2, 'mean') so3_geodesic_eps_f32 (workspace: one launch)", lambda i: lib.so3_geodesic_eps_f32(p(r[i % NB]), p(rt[i % NB]), None, p(ls), p(lm), 1, ctypes.c_float(1e-7), p(ws), n, st), 72 * n)
    timeit("K4' geodesic(R1, R2, 'mean') so3_geodesic_eps_f32 (no workspace: memset + kernel + mean)", lambda i: lib.so3_geodesic_eps_f32(p(r[i % NB]), p(rt[i % NB]), None, p(ls), p(lm), 1, ctypes.c_float(1e-7), None, n, st), 72 * n)
    del r64, g64
    # K4b: the metrics' backward (round 6): dR1 / dR2 of geodesic(..., 'mean') from a 0-dim upstream gradient, of the per-row form from a
    # per-row one, and angle_error's float64 spelling
    RAD, GS, F64 = _lib.RADIANS, _lib.GRAD_SCALAR, _lib.F64_MATH
    one32, one64 = torch.ones(1, device=dev), torch.ones(1, dtype=torch.float64, device=dev)
    w32, w64 = torch.randn(n, device=dev), torch.randn(n, dtype=torch.float64, device=dev)
    kb = lib.so3_angle_bwd_f32
    D = ctypes.c_double
    timeit("K4b so3_angle_bwd_f32 geodesic mean: dR1 + dR2", lambda i: kb(p(r[i % NB]), p(rt[i % NB]), p(one32), D(n), D(1e-7), RAD | GS, p(dm[i % NB]), p(g[i % NB]), n, st), 144 * n)
    timeit("K4b so3_angle_bwd_f32 geodesic mean: dR1 alone", lambda i: kb(p(r[i % NB]), p(rt[i % NB]), p(one32), D(n), D(1e-7), RAD | GS, p(dm[i % NB]), None, n, st), 108 * n)
    timeit("K4b so3_angle_bwd_f32 per-row upstream gradient: dR1 alone", lambda i: kb(p(r[i % NB]), p(rt[i % NB]), p(w32), D(1), D(0), RAD, p(dm[i % NB]), None, n, st), 112 * n)
    timeit("K4b so3_angle_bwd_f32 angle_error.mean() (float64 math): dR1 alone", lambda i: kb(p(r[i % NB]), p(rt[i % NB]), p(one64), D(n), D(0), GS | F64, p(dm[i % NB]), None, n, st), 108 * n)
    timeit("K4b so3_angle_bwd_f32 angle_error per-row float64 gradient: dR1 + dR2", lambda i: kb(p(r[i % NB]), p(rt[i % NB]), p(w64), D(1), D(0), F64, p(dm[i % NB]), p(g[i % NB]), n, st), 152 * n)
    timeit("K4' so3_geodesic_f32", lambda i: lib.so3_geodesic_f32(p(r[i % NB]), p(rt[i % NB]), p(th), n, st), 76 * n)
    print("--- next rows (f1, f2, f3) at 1M rows ---")
    x6 = [torch.randn(n, 6, device=dev) for _ in range(NB)]
    d6 = torch.empty(n, 6, device=dev)
    timeit("f2 so3_ortho6d_fwd_f32", lambda i: lib.so3_ortho6d_fwd_f32(p(x6[i % NB]), p(r[i % NB]), n, st), 60 * n)
    timeit("f2 so3_ortho6d_bwd_f32", lambda i: lib.so3_ortho6d_bwd_f32(p(x6[i % NB]), p(g[i % NB]), p(d6), n, st), 84 * n)
    del x6, d6
    for name, w in (("quat", 4), ("euler", 3), ("ortho5d", 5), ("expmap", 3)):
        xh = [torch.randn(n, w, device=dev) for _ in range(NB)]
        dh = torch.empty(n, w, device=dev)
        fwd, bwd = getattr(lib, "so3_%s_fwd_f32" % name), getattr(lib, "so3_%s_bwd_f32" % name)
        timeit("f5 so3_%s_fwd_f32" % name, lambda i: fwd(p(xh[i % NB]), p(r[i % NB]), n, st), (4 * w + 36) * n)
        timeit("f5 so3_%s_bwd_f32" % name, lambda i: bwd(p(xh[i % NB]), p(g[i % NB]), p(dh), n, st), (8 * w + 36) * n)
        del xh, dh
    o12 = [torch.randn(n, 12, device=dev) for _ in range(3)]
    ti = [torch.eye(4, device=dev).repeat(n, 1, 1).contiguous() + 0.1 * torch.randn(n, 4, 4, device=dev) for _ in range(3)]
    tp = torch.empty(n, 16, device=dev); g16 = torch.randn(n, 16, device=dev); do12 = torch.empty(n, 12, device=dev)
    fx = ctypes.c_float(444.444)
    timeit("f1 so3_se3_update_f32", lambda i: lib.so3_se3_update_f32(p(o12[i % 3]), p(ti[i % 3]), p(tp), fx, fx, n, st), 176 * n)
    timeit("f1 so3_se3_update_bwd_f32", lambda i: lib.so3_se3_update_bwd_f32(p(o12[i % 3]), p(ti[i % 3]), p(g16), p(do12), fx, fx, n, st), 224 * n)
    del o12, ti, tp, g16, do12
    cls = torch.randint(0, 10, (n,), device=dev, dtype=torch.int32)
    lib.so3_angle_error_v2(p(r[0]), p(rt[0]), p(deg), None, p(fl), None, 0, n, st)
    stats = torch.empty(10, 8, dtype=torch.float64, device=dev)
    work = torch.zeros(lib.so3_angle_stats_workspace_bytes(), dtype=torch.uint8, device=dev)     # zero-filled once
    timeit("f3 so3_angle_stats (10 classes, exact median)", lambda i: lib.so3_angle_stats(p(deg), p(cls), 10, p(stats), p(work), n, st), 12 * n, iters=20)
    timeit("f3 so3_angle_stats (one class: the whole batch's median)", lambda i: lib.so3_angle_stats(p(deg), None, 1, p(stats), p(work), n, st), 8 * n, iters=20)
    del x, xb, g, r, dm, dmb, rt
    torch.cuda.empty_cache()
    print("--- config #3: 65536 clouds x 1024 points ---")
    b, npts = 65536, 1024
    pc = [torch.rand(b, npts, 3, device=dev) - 0.5 for _ in range(2)]
    qc = [torch.rand(b, npts, 3, device=dev) - 0.5 for _ in range(2)]
    rk = torch.empty(b, 9, device=dev)
    timeit("K5 so3_kabsch_f32", lambda i: lib.so3_kabsch_f32(p(pc[i % 2]), p(qc[i % 2]), p(rk), None, b, npts, st), b * (2 * npts * 12 + 36), iters=10, warm=2)
    rg = rr.get_sampled_rotation_matrices_by_axisAngle(b, dev).reshape(b, 9).contiguous()
    timeit("f4 so3_kabsch_synth_f32 (sigma=0: P only)", lambda i: lib.so3_kabsch_synth_f32(p(pc[i % 2]), p(rg), ctypes.c_float(0.0), 1, p(rk), None, b, npts, st), b * (npts * 12 + 72), iters=10, warm=2)
    timeit("f4 so3_kabsch_synth_f32 (sigma=0.01, device RNG)", lambda i: lib.so3_kabsch_synth_f32(p(pc[i % 2]), p(rg), ctypes.c_float(0.01), 1, p(rk), None, b, npts, st), b * (npts * 12 + 72), iters=10, warm=2)
    del qc
    tg = torch.eye(4, device=dev).repeat(b, 1, 1).contiguous(); tg[:, :3, :3] = rg.view(b, 3, 3); tg[:, :3, 3] = torch.randn(b, 3, device=dev)
    tq = tg.clone(); tq[:, :3, :3] = rr.get_sampled_rotation_matrices_by_axisAngle(b, dev); tq[:, :3, 3] += 0.1 * torch.randn(b, 3, device=dev)
    dtq = torch.empty(b, 16, device=dev); l3 = torch.empty(3, dtype=torch.float64, device=dev)
    sc = ctypes.c_float(1.0 / b)
    timeit("f6 so3_add_l1_f32 (loss + dTpred)", lambda i: lib.so3_add_l1_f32(p(tg), p(tq), p(pc[i % 2]), None, p(l3), p(dtq), sc, b, npts, st), b * (npts * 12 + 192), iters=10, warm=2)
    timeit("f6 so3_add_l1_disentangled_f32 (loss + dTpred)", lambda i: lib.so3_add_l1_disentangled_f32(p(tq), p(tg), p(pc[i % 2]), p(l3), p(dtq), sc, b, npts, st), b * (npts * 12 + 192), iters=10, warm=2)
    qo = torch.empty(b, npts, 3, device=dev)
    timeit("a7 so3_rotate_clouds_f32", lambda i: lib.so3_rotate_clouds_f32(p(pc[i % 2]), p(rg), p(qo), 0, b, npts, st), b * (npts * 24 + 36), iters=10, warm=2)
    timeit("a7 so3_rotate_clouds_f32 (transposed out)", lambda i: lib.so3_rotate_clouds_f32(p(pc[i % 2]), p(rg), p(qo), 1, b, npts, st), b * (npts * 24 + 36), iters=10, warm=2)
    timeit("a7 so3_pc_normalize_f32", lambda i: lib.so3_pc_normalize_f32(p(pc[i % 2]), p(qo), None, None, b, npts, st), b * (npts * 24), iters=10, warm=2)
    del pc, tg, tq, dtq, qo
    torch.cuda.empty_cache()
    print("--- config #4: B = 512, bf16 storage, fused head + loss + backward ---")
    b = 512
    x4 = torch.randn(b, 9, device=dev).bfloat16(); r4 = torch.empty(b, 9, device=dev); d4 = torch.empty(b, 9, device=dev, dtype=torch.bfloat16)
    t4 = rr.symmetric_orthogonalization(torch.randn(b, 9, device=dev))
    timeit("K3 so3_frob_fwd_bwd_bf16 (B=512)", lambda i: lib.so3_frob_fwd_bwd_v2_bf16(p(x4), p(t4), p(r4), p(d4), p(ls), None, None, 0, b, st), b * (18 + 36 + 36 + 18), iters=300)
    g4 = torch.empty(b, 9, device=dev)
    timeit("K3' so3_frob_loss_f32 (B=512, loss + dRpred: one launch)", lambda i: lib.so3_frob_loss_v2_f32(p(r4), p(t4), p(g4), p(ls), None, None, 0, b, st), b * 108, iters=300)
    x4f = x4.float(); sc4 = torch.empty(2, dtype=torch.float64, device=dev); fl4 = torch.zeros(1, dtype=torch.int32, device=dev)
    timeit("K4 so3_angle_error (B=512, sum,count + flag: one launch)", lambda i: lib.so3_angle_error_v2(p(r4), p(t4), None, p(sc4), p(fl4), None, 0, b, st), b * 72, iters=300)
    timeit("K1+K4 so3_project_angle_error_f32 (B=512, sum,count + flag)", lambda i: lib.so3_project_angle_error_v2_f32(p(x4f), p(t4), None, None, p(sc4), p(fl4), None, 4, b, st), b * 72, iters=300)
    print("--- config #1: B = 256 ---")
    x1 = torch.randn(256, 9, device=dev); r1 = torch.empty(256, 9, device=dev)
    timeit("K1 so3_project_fwd_f32 (B=256)", lambda i: lib.so3_project_fwd_f32(p(x1), p(r1), None, 256, st), 256 * 72, iters=300)


if __name__ == "__main__":
    main()
