"""Adversarial search on the fast path's certificate, judged by the DEVICE's own arithmetic (v_rsq / v_rcp / v_cos at 1 ulp,
fused contractions as hipcc emits them) -- not by the host model's libm.

The quaternion fast path (csrc/so3_device.h, quat_rotation) keeps a row only when its own tests certify it (gap product,
Rayleigh move, residual, curvature, scale window); everything else goes to the Jacobi path.  What could go wrong silently is an
ACCEPTED row that is inaccurate.  An evolutionary loop looks for one: 1M candidates per generation, scored by the error of the
accepted rows in the measure the kernel is judged by (|dR| gap / s1), the worst ones bred (rescaled, perturbed at every
relative size, blended, rotated) into the next generation, beside fresh draws from the families that found bugs in round 2
(entries near the window's edges, exact double roots, near-reflections, rank deficiency).  The bulk reference is the float64
device kernel; the worst candidates of every generation are re-judged against float64 LAPACK on the host (independent)."""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N = 1_000_000
GENERATIONS = 20                      # 2e7 rows
# max accepted |dR| gap / s1.  The build's own bar (north_star asks ||R^T R - I|| < 1e-5 and the mean angle to 1e-4 degrees: both met with an order of
# magnitude to spare).  Stated with headroom since round 6: the worst row of SIXTY searches reads 1.90e-6 (profiles/r0N_search_seeds.txt), and a bound
# of 2e-6 asserted on three seeds of a stochastic search that reruns every round had 5 % of margin -- a red run would have been a margin, not a bug.
BOUND = 2.5e-6


def _haar(n, gen):
    q = torch.randn(n, 4, device=DEV, generator=gen, dtype=torch.float64)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=1).view(n, 3, 3)


def _with_singular_values(s, gen):
    u, v = _haar(len(s), gen), _haar(len(s), gen)
    return (u * s.unsqueeze(1)) @ v.transpose(1, 2)


def _seeds(n, gen):
    """Fresh draws from the families that broke (or nearly broke) the certificate before."""
    k = n // 8
    u = lambda m, lo, hi: torch.rand(m, device=DEV, generator=gen, dtype=torch.float64) * (hi - lo) + lo
    g = lambda m: torch.randn(m, 3, 3, device=DEV, generator=gen, dtype=torch.float64)
    parts = [
        g(k),                                                                                  # the benchmark's distribution
        g(k) * (2.0 ** u(k, 13.0, 19.0)).view(-1, 1, 1),                                       # around the window's upper edge (2^34 on |M|^2)
        g(k) * (2.0 ** u(k, -19.0, -12.0)).view(-1, 1, 1),                                     # ... and its lower edge (2^-28)
        torch.randint(-2, 3, (k, 3, 3), device=DEV, generator=gen).double(),                   # small integers: exact double roots, rank deficiency
        -_haar(k, gen) + g(k) * (10.0 ** u(k, -7.0, -1.0)).view(-1, 1, 1),                     # near-reflections at every distance
    ]
    s = torch.stack([u(k, 0.5, 2.0), u(k, 0.1, 1.0), torch.zeros(k, device=DEV, dtype=torch.float64)], 1)
    s[:, 2] = -s[:, 1] * (1 - 10.0 ** u(k, -8.0, -1.0))                                        # s2 ~ s3, det < 0: the gap at every size
    parts.append(_with_singular_values(s, gen))
    s = torch.stack([u(k, 0.5, 2.0), 10.0 ** u(k, -7.0, 0.0), torch.zeros(k, device=DEV, dtype=torch.float64)], 1)
    s[:, 2] = s[:, 1] * 10.0 ** u(k, -7.0, 0.0) * torch.sign(u(k, -1.0, 1.0))                  # small s2, s3 of either sign
    parts.append(_with_singular_values(s, gen))
    rest = n - sum(len(p) for p in parts)
    parts.append(g(rest) * g(rest).abs().clamp_min(1e-3))                                      # heavy-tailed entries
    return torch.cat(parts).reshape(n, 9)


def _breed(parents, n, gen):
    """n children of the parent rows (float64, (P,9)): rescale, perturb at a random relative size, blend, rotate."""
    idx = torch.randint(0, len(parents), (n,), device=DEV, generator=gen)
    child = parents[idx].clone().view(n, 3, 3)
    kind = torch.randint(0, 5, (n,), device=DEV, generator=gen)
    rel = 10.0 ** (torch.rand(n, device=DEV, generator=gen, dtype=torch.float64) * 8.0 - 8.5)   # 3e-9 .. 3e-1
    scale = child.abs().amax(dim=(1, 2), keepdim=True).clamp_min(1e-300)
    noise = torch.randn(n, 3, 3, device=DEV, generator=gen, dtype=torch.float64) * scale * rel.view(-1, 1, 1)
    child = torch.where((kind <= 1).view(-1, 1, 1), child + noise, child)                      # additive noise
    pw = 2.0 ** torch.randint(-3, 4, (n,), device=DEV, generator=gen).double()
    child = torch.where((kind == 2).view(-1, 1, 1), child * pw.view(-1, 1, 1), child)          # exact rescale
    mate = parents[torch.randint(0, len(parents), (n,), device=DEV, generator=gen)].view(n, 3, 3)
    mix = torch.rand(n, device=DEV, generator=gen, dtype=torch.float64).view(-1, 1, 1)
    child = torch.where((kind == 3).view(-1, 1, 1), mix * child + (1 - mix) * mate, child)     # blend
    rot = _haar(n, gen)
    child = torch.where((kind == 4).view(-1, 1, 1), rot @ child, child)                        # same singular values, new orientation
    return child.reshape(n, 9)


def _sym_eigs(s):
    """Eigenvalues (ascending) of symmetric 3x3 blocks (n,3,3), float64, trigonometric closed form -- for SCORING only."""
    q = (s[:, 0, 0] + s[:, 1, 1] + s[:, 2, 2]) / 3
    d = s - q.view(-1, 1, 1) * torch.eye(3, device=s.device, dtype=s.dtype)
    p = (d * d).sum((1, 2)).div(6).sqrt().clamp_min(1e-300)
    b = d / p.view(-1, 1, 1)
    phi = torch.acos((torch.linalg.det(b) / 2).clamp(-1, 1)) / 3
    e1 = q + 2 * p * torch.cos(phi)
    e3 = q + 2 * p * torch.cos(phi + 2 * np.pi / 3)
    return torch.stack([e3, 3 * q - e1 - e3, e1], 1)


def _lapack(m32):
    """float64 LAPACK on the host: R_ref, s, flip for a handful of rows (the independent judge)."""
    m = m32.astype(np.float64).reshape(-1, 3, 3)
    u, s, vt = np.linalg.svd(m)
    d = np.linalg.det(u @ vt)
    vt = vt.copy()
    vt[:, 2, :] *= d[:, None]
    return u @ vt, s, d < 0


@pytest.mark.parametrize("seed", [2025, 42, 31337])          # (tools/search_seeds.py: any seeds, any build; ten of them read 1.42-1.75e-6)
def test_adversarial_search_finds_no_accepted_row_beyond_the_bound(seed):
    from poseestimation_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device=DEV).manual_seed(int(os.environ.get("SO3_SEARCH_SEED", seed)))      # (another seed: another search, same bound)
    pop = _seeds(N, gen)
    worst_overall, accepted_total, hard_total, flips_checked = 0.0, 0, 0, 0
    history = []
    for generation in range(GENERATIONS):
        x32 = pop.float()
        finite = torch.isfinite(x32).all(1)
        x32 = torch.where(finite.view(-1, 1), x32, torch.zeros_like(x32)).contiguous()
        x64 = x32.double()
        r32 = torch.empty(N, 9, device=DEV)
        hard = torch.empty(N, dtype=torch.uint8, device=DEV)
        flip = torch.empty(N, dtype=torch.uint8, device=DEV)
        rk1 = torch.empty(N, 9, device=DEV)
        r64 = torch.empty(N, 9, device=DEV, dtype=torch.float64)
        assert lib.so3_project_fwd_diag_f32(x32.data_ptr(), r32.data_ptr(), hard.data_ptr(), N, st) == 0
        assert lib.so3_project_fwd_f32(x32.data_ptr(), rk1.data_ptr(), flip.data_ptr(), N, st) == 0       # the product kernel (packed engine)
        assert lib.so3_project_fwd_f64(x64.data_ptr(), r64.data_ptr(), None, N, st) == 0
        assert torch.equal(rk1, r32)                          # the diagnostic twin IS the product's arithmetic, bit for bit
        acc = hard == 0
        # every accepted row is a rotation
        rr_ = r32.double().view(N, 3, 3)
        orth = (rr_.transpose(1, 2) @ rr_ - torch.eye(3, device=DEV, dtype=torch.float64)).abs().amax((1, 2))
        assert orth[acc].max().item() < 3e-6
        # score: |dR| gap / s1 with gap, s1 from S = R64^T M (symmetric, eigenvalues s3', s2, s1)
        m = x64.view(N, 3, 3)
        s_mat = r64.view(N, 3, 3).transpose(1, 2) @ m
        eig = _sym_eigs(0.5 * (s_mat + s_mat.transpose(1, 2)))
        s1 = eig[:, 2].clamp_min(1e-300)
        gap = (eig[:, 0] + eig[:, 1]).clamp_min(0)
        err = (r32.double() - r64).abs().amax(1)
        score = torch.where(acc, err * gap / s1, torch.zeros_like(err))
        score = torch.where(torch.isfinite(score), score, torch.zeros_like(score))
        top = torch.topk(score, 2000).indices
        # the independent judge on the worst candidates: float64 LAPACK on the host
        rows = x32[top].cpu().numpy()
        r_ref, sv, flip_ref = _lapack(rows)
        det_sign_safe = np.abs(np.linalg.det(rows.astype(np.float64).reshape(-1, 3, 3))) > 1e-12 * sv[:, 0] ** 3
        gap_ref = np.where(flip_ref, sv[:, 1] - sv[:, 2], sv[:, 1] + sv[:, 2])
        err_ref = np.abs(r32[top].cpu().numpy().reshape(-1, 3, 3) - r_ref).reshape(len(top), -1).max(1)
        judged = err_ref * gap_ref / np.maximum(sv[:, 0], 1e-300)
        worst = float(judged.max())
        worst_overall = max(worst_overall, worst)
        history.append((generation, int(acc.sum().item()), worst, float(score.max().item())))
        assert worst <= BOUND, (generation, worst, rows[int(judged.argmax())].tolist())
        # flip flags, bit-exact wherever the sign of det is decided in float64
        det64 = torch.linalg.det(m)
        mag = m.abs().amax((1, 2)) ** 3
        decided = det64.abs() > 1e-12 * mag
        assert torch.equal(flip[decided] != 0, det64[decided] < 0)
        flips_checked += int(decided.sum().item())
        assert np.array_equal(flip[top].cpu().numpy().astype(bool)[det_sign_safe], flip_ref[det_sign_safe])
        accepted_total += int(acc.sum().item())
        hard_total += int((~acc).sum().item())
        # next generation: the worst accepted rows breed; a quarter of the population is fresh seed material
        parents = x64[torch.topk(score, 10_000).indices]
        pop = torch.cat([_breed(parents, N - N // 4, gen), _seeds(N // 4, gen)])
    assert accepted_total + hard_total == GENERATIONS * N >= 20_000_000
    assert accepted_total > 5_000_000 and flips_checked > 10_000_000          # the search did exercise the fast path
    print("\\nadversarial search: %d rows, %d accepted, worst accepted |dR| gap/s1 = %.3g (bound %.2g); per generation "
          "(accepted, LAPACK-judged worst, device-scored worst): %s"
          % (GENERATIONS * N, accepted_total, worst_overall, BOUND, [(a, "%.2g" % w, "%.2g" % s) for _, a, w, s in history]))
