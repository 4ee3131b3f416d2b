"""Numerical prototype (numpy, fp32 emulation) of the per-lane 3x3 signed-SVD used by the HIP kernels.

Shipped schedule (csrc/so3_device.h): 3 fixed sweeps, then one more for the rows whose (0,1) residual exceeds
0.7e-5 (in the kernel: for the whole wave).  This script is the design study behind that choice.

Design study only -- not shipped, not imported by the package.  It answers: how many one-sided
(Hestenes) Jacobi sweeps does fp32 need so that R = U'V^T matches LAPACK's
U diag(1,1,det(UV^T)) V^T on Gaussian 3x3 input?
"""
import sys
import numpy as np

f32 = np.float32


def _dot(x, y):
    return (x[:, 0] * y[:, 0] + x[:, 1] * y[:, 1] + x[:, 2] * y[:, 2]).astype(f32)


def _normalize(x):
    n2 = _dot(x, x)
    return (x * (f32(1) / np.sqrt(n2, dtype=f32))[:, None]).astype(f32)


def _gs(x1, x2):
    u1 = _normalize(x1)
    w = (x2 - _dot(u1, x2)[:, None] * u1).astype(f32)
    u2 = _normalize(w)
    u3 = np.cross(u1, u2).astype(f32)
    return u1, u2, u3


def signed_svd_hestenes(M, sweeps, accumulate_v=True, track_norms=False):
    """M: (B,3,3) float32.  Returns R (B,3,3).  All arithmetic in float32.

    One-sided Jacobi: A <- A J (J plane rotations) until the columns of A are orthogonal;
    A = M V = U' S'.  R = U' V^T with U' = [u1, u2, u1 x u2] (no explicit det flip needed).
    """
    M = M.astype(f32)
    A = [M[:, :, k].copy() for k in range(3)]   # columns a_k (B,3)
    B = M.shape[0]
    V = [np.zeros((B, 3), f32) for _ in range(3)]
    for k in range(3):
        V[k][:, k] = 1
    tiny = f32(1e-37)
    n = [_dot(A[k], A[k]) for k in range(3)]
    for _ in range(sweeps):
        for (p, q) in ((0, 1), (0, 2), (1, 2)):
            if track_norms:
                al, be = n[p], n[q]
            else:
                al, be = _dot(A[p], A[p]), _dot(A[q], A[q])
            ga = _dot(A[p], A[q])
            e = f32(0.5) * (al - be)
            h = np.sqrt(e * e + ga * ga, dtype=f32)
            ae = np.abs(e) + h
            w = np.maximum(f32(2) * h * ae, tiny)
            rw = (f32(1) / np.sqrt(w, dtype=f32)).astype(f32)
            c = np.where(h > 0, ae * rw, f32(1)).astype(f32)
            s = (ga * rw).astype(f32)
            s = np.where(e < 0, -s, s).astype(f32)
            # a_p' = c a_p + s a_q ; a_q' = c a_q - s a_p
            c_, s_ = c[:, None], s[:, None]
            A[p], A[q] = (c_ * A[p] + s_ * A[q]).astype(f32), (c_ * A[q] - s_ * A[p]).astype(f32)
            if accumulate_v:
                V[p], V[q] = (c_ * V[p] + s_ * V[q]).astype(f32), (c_ * V[q] - s_ * V[p]).astype(f32)
            m = f32(0.5) * (al + be)
            big, small = m + h, np.maximum(m - h, f32(0))
            n[p] = np.where(e < 0, small, big).astype(f32)
            n[q] = np.where(e < 0, big, small).astype(f32)
    n = [_dot(A[k], A[k]) for k in range(3)]

    def cswap(i, j):
        sw = n[i] < n[j]
        s = sw[:, None]
        A[i], A[j] = np.where(s, A[j], A[i]), np.where(s, -A[i], A[j])
        V[i], V[j] = np.where(s, V[j], V[i]), np.where(s, -V[i], V[j])
        n[i], n[j] = np.where(sw, n[j], n[i]), np.where(sw, n[i], n[j])
    cswap(0, 1); cswap(0, 2); cswap(1, 2)
    u1, u2, u3 = _gs(A[0], A[1])
    if accumulate_v:
        v1, v2, v3 = _gs(V[0], V[1])
    else:
        Mt = M.transpose(0, 2, 1)
        v1, v2, v3 = _gs(np.einsum('bij,bj->bi', Mt, u1).astype(f32), np.einsum('bij,bj->bi', Mt, u2).astype(f32))
    R = (u1[:, :, None] * v1[:, None, :] + u2[:, :, None] * v2[:, None, :] + u3[:, :, None] * v3[:, None, :])
    return R.astype(f32)


def ref_proj(M):
    U, S, Vt = np.linalg.svd(M)
    d = np.linalg.det(U @ Vt)
    Vt = Vt.copy()
    Vt[:, 2, :] *= d[:, None]
    return U @ Vt, S, d


def angle_deg(R1, R2):
    tr = np.einsum('bji,bji->b', R1.astype(np.float64), R2.astype(np.float64))
    return np.degrees(np.arccos(np.clip((tr - 1) / 2, -1, 1)))


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    rng = np.random.default_rng(0)
    M = rng.standard_normal((B, 3, 3)).astype(f32)
    T = ref_proj(rng.standard_normal((B, 3, 3)))[0]
    R64, S, d = ref_proj(M.astype(np.float64))
    R32, _, _ = ref_proj(M)
    gap = np.where(d < 0, S[:, 1] - S[:, 2], S[:, 1] + S[:, 2]) / S[:, 0]
    ok = gap > 1e-3
    print("LAPACK f32 vs f64: max|dR| %.3e  (well-cond rows %.3e)  mean-angle delta %.3e" % (
        np.abs(R32 - R64).max(), np.abs(R32 - R64)[ok].max(), angle_deg(R32, T).mean() - angle_deg(R64, T).mean()))
    import itertools
    for (accv, track), sweeps in itertools.product(((True, False), (False, False), (True, True), (False, True)), (3, 4, 5, 6)):
        R = signed_svd_hestenes(M, sweeps, accv, track)
        print("accV=%d track=%d" % (accv, track), end=" ")
        err = np.abs(R - R64).reshape(B, -1).max(1)
        orth = np.linalg.norm(np.einsum('bji,bjk->bik', R, R) - np.eye(3), axis=(1, 2))
        print("sweeps %d: max|dR| %.3e  wellcond max %.3e  p99.9 %.3e  median %.3e  orth max %.3e  d(mean angle) %.3e deg" % (
            sweeps, err.max(), err[ok].max(), np.quantile(err, 0.999), np.median(err), orth.max(),
            angle_deg(R, T).mean() - angle_deg(R64, T).mean()))
