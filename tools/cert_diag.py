#!/usr/bin/env python3
"""Which rows carry the largest accepted |dR| gap/s1, per build: the certificate search's generation-0 population through two libraries.
usage: cert_diag.py a.so b.so"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import test_gpu_certificate_search as cs

N = cs.N
gen = torch.Generator(device=cs.DEV).manual_seed(2025)
pop = cs._seeds(N, gen)
x32 = pop.float().contiguous()
x64 = x32.double()
P = ctypes.c_void_p
st = P(torch.cuda.current_stream().cuda_stream)
res = {}
for path in sys.argv[1:]:
    lib = ctypes.CDLL(path)
    r32 = torch.empty(N, 9, device=cs.DEV); hard = torch.empty(N, dtype=torch.uint8, device=cs.DEV)
    r64 = torch.empty(N, 9, device=cs.DEV, dtype=torch.float64)
    assert lib.so3_project_fwd_diag_f32(P(x32.data_ptr()), P(r32.data_ptr()), P(hard.data_ptr()), ctypes.c_int64(N), st) == 0
    assert lib.so3_project_fwd_f64(P(x64.data_ptr()), P(r64.data_ptr()), None, ctypes.c_int64(N), st) == 0
    m = x64.view(N, 3, 3)
    s_mat = r64.view(N, 3, 3).transpose(1, 2) @ m
    eig = cs._sym_eigs(0.5 * (s_mat + s_mat.transpose(1, 2)))
    s1 = eig[:, 2].clamp_min(1e-300); gap = (eig[:, 0] + eig[:, 1]).clamp_min(0)
    err = (r32.double() - r64).abs().amax(1)
    score = torch.where(hard == 0, err * gap / s1, torch.zeros_like(err))
    score = torch.where(torch.isfinite(score), score, torch.zeros_like(score))
    res[path] = (score, hard, eig, err)
names = list(res)
k = N // 8
fam = ["gaussian", "upper edge", "lower edge", "small integers", "near-reflections", "s2~s3 det<0", "small s2", "heavy-tailed"]
for name in names:
    score, hard, eig, err = res[name]
    print("==", os.path.basename(name), "worst %.3g" % score.max().item())
    for f in range(8):
        sl = slice(f * k, (f + 1) * k) if f < 7 else slice(7 * k, N)
        print("   %-18s worst %.3g  p99.99 %.3g  accepted %d" % (fam[f], score[sl].max().item(), torch.quantile(score[sl][:200000].float(), 0.9999).item(), int((hard[sl] == 0).sum())))
a, b = res[names[0]], res[names[-1]]
top = torch.topk(b[0], 12).indices
for i in top.tolist():
    e = b[2][i]
    print("row %7d fam %-16s  score a %.3g b %.3g  hard a %d b %d  eig(s3',s2,s1) %.4g %.4g %.4g  lam/s1 %.3f  err b %.3g  scale |M| %.3g"
          % (i, fam[min(i // k, 7)], a[0][i].item(), b[0][i].item(), a[1][i].item(), b[1][i].item(), e[0].item(), e[1].item(), e[2].item(),
             ((e[0] + e[1] + e[2]) / e[2]).item(), b[3][i].item(), x64[i].norm().item()))
