"""Dump the worst accepted rows of generation 0 of tests/test_gpu_certificate_search.py with everything known about them."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import test_gpu_certificate_search as T
from poseestimation_amd import _lib
from oracle import kernel_model as km
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device=T.DEV).manual_seed(2025)
N = T.N
pop = T._seeds(N, gen)
x32 = pop.float().contiguous()
r32 = torch.empty(N, 9, device=T.DEV); hard = torch.empty(N, dtype=torch.uint8, device=T.DEV)
r64 = torch.empty(N, 9, device=T.DEV, dtype=torch.float64)
lib.so3_project_fwd_diag_f32(x32.data_ptr(), r32.data_ptr(), hard.data_ptr(), N, st)
lib.so3_project_fwd_f64(x32.double().data_ptr(), r64.data_ptr(), None, N, st)
acc = hard == 0
m = x32.double().view(N, 3, 3)
s_mat = r64.view(N, 3, 3).transpose(1, 2) @ m
eig = T._sym_eigs(0.5 * (s_mat + s_mat.transpose(1, 2)))
score = torch.where(acc, (r32.double() - r64).abs().amax(1) * (eig[:, 0] + eig[:, 1]).clamp_min(0) / eig[:, 2].clamp_min(1e-300), torch.zeros(N, device=T.DEV, dtype=torch.float64))
score = torch.where(torch.isfinite(score), score, torch.zeros_like(score))
top = torch.topk(score, 4000).indices
rows = x32[top].cpu().numpy()
r_ref, sv, flip_ref = T._lapack(rows)
gap_ref = np.where(flip_ref, sv[:, 1] - sv[:, 2], sv[:, 1] + sv[:, 2])
got = r32[top].cpu().numpy().reshape(-1, 3, 3)
err_ref = np.abs(got - r_ref).reshape(len(top), -1).max(1)
judged = err_ref * gap_ref / sv[:, 0]
order = np.argsort(-judged)[:12]
rm, hm = km.project_quat(rows)
rj = km.project_jacobi(rows)
np.set_printoptions(precision=9, linewidth=200)
print("family of index: i // (N//8):")
for i in order:
    print("idx %d fam %d judged %.3g err %.3g gap/s1 %.3g s %s flip %s | model hard %s model err %.3g jacobi err %.3g | dev-vs-model %.3g" % (
        int(top[i]), int(top[i]) // (N // 8), judged[i], err_ref[i], gap_ref[i] / sv[i, 0], sv[i], flip_ref[i], hm[i],
        np.abs(rm[i] - r_ref[i]).max(), np.abs(rj[i] - r_ref[i]).max(), np.abs(rm[i] - got[i]).max()))
    print("   row", rows[i].tolist())
print("judged > 2e-6:", (judged > 2e-6).sum(), "of", len(judged), " families:", np.bincount(top.cpu().numpy()[judged > 2e-6] // (N // 8), minlength=8))
