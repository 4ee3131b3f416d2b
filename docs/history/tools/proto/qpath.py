"""Prototype (numpy float32): SO(3) projection through the dominant eigenvector of Davenport's 4x4 matrix.
Experiment for K1: cheaper than 9 Jacobi rotations?  Measures accuracy against float64 SVD."""
import numpy as np, sys
f32 = np.float32

def ref_f64(M):
    M = M.astype(np.float64)
    U, S, Vt = np.linalg.svd(M)
    d = np.sign(np.linalg.det(U @ Vt))
    d[d == 0] = 1
    D = np.zeros_like(M); D[:, 0, 0] = 1; D[:, 1, 1] = 1; D[:, 2, 2] = d
    return U @ D @ Vt, S, d

def qpath(M, newton_iters=5, refine=1, stats=None):
    M = M.astype(f32)
    n = len(M)
    # prescale by max abs (power of two)
    mx = np.abs(M).reshape(n, -1).max(1)
    e = np.where(mx > 0, np.floor(np.log2(np.maximum(mx, 1e-45))) + 1, 0)
    M = (M * (f32(2.0) ** (-e)).astype(f32)[:, None, None]).astype(f32)
    m = lambda i, j: M[:, i, j]
    tr = m(0, 0) + m(1, 1) + m(2, 2)
    kd = np.stack([tr, f32(2) * m(0, 0) - tr, f32(2) * m(1, 1) - tr, f32(2) * m(2, 2) - tr], 1)   # ww, xx, yy, zz
    j = kd.argmax(1)                     # 0: no flip; 1: D=diag(1,-1,-1); 2: diag(-1,1,-1); 3: diag(-1,-1,1)
    sg = np.ones((n, 3), f32)
    sg[j == 1] = [1, -1, -1]; sg[j == 2] = [-1, 1, -1]; sg[j == 3] = [-1, -1, 1]
    M = M * sg[:, None, :]
    tr = m(0, 0) + m(1, 1) + m(2, 2)
    # K in order (x, y, z, w)
    Kxx = f32(2) * m(0, 0) - tr; Kyy = f32(2) * m(1, 1) - tr; Kzz = f32(2) * m(2, 2) - tr; Kww = tr
    Kxy = m(0, 1) + m(1, 0); Kxz = m(0, 2) + m(2, 0); Kyz = m(1, 2) + m(2, 1)
    Kxw = m(2, 1) - m(1, 2); Kyw = m(0, 2) - m(2, 0); Kzw = m(1, 0) - m(0, 1)
    # characteristic polynomial  l^4 + c2 l^2 + c1 l + c0
    fro2 = (M * M).reshape(n, -1).sum(1, dtype=f32)
    detM = (m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) - m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0))
            + m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0))).astype(f32)
    c2 = f32(-2) * fro2; c1 = f32(-8) * detM
    # det K by complementary 2x2 minors of rows (x,y) and (z,w)
    a = [Kxx, Kxy, Kxz, Kxw]; b = [Kxy, Kyy, Kyz, Kyw]; c = [Kxz, Kyz, Kzz, Kzw]; d = [Kxw, Kyw, Kzw, Kww]
    def mn(p, q, i, k): return p[i] * q[k] - p[k] * q[i]
    c0 = (mn(a, b, 0, 1) * mn(c, d, 2, 3) - mn(a, b, 0, 2) * mn(c, d, 1, 3) + mn(a, b, 0, 3) * mn(c, d, 1, 2)
          + mn(a, b, 1, 2) * mn(c, d, 0, 3) - mn(a, b, 1, 3) * mn(c, d, 0, 2) + mn(a, b, 2, 3) * mn(c, d, 0, 1)).astype(f32)
    lam = np.sqrt(f32(3) * fro2).astype(f32)
    for it in range(newton_iters):
        l2 = lam * lam
        P = ((l2 + c2) * lam + c1) * lam + c0
        dP = (f32(4) * l2 + f32(2) * c2) * lam + c1
        lam = (lam - P / dP).astype(f32)
    def nullvec(lam):
        # N = lam I - K (PSD up to rounding), LDL^T without pivoting in order x,y,z,w; null vector = L^-T e4
        n11 = lam - Kxx; n21 = -Kxy; n31 = -Kxz; n41 = -Kxw
        n22 = lam - Kyy; n32 = -Kyz; n42 = -Kyw; n33 = lam - Kzz; n43 = -Kzw; n44 = lam - Kww
        i1 = f32(1) / n11
        l21 = n21 * i1; l31 = n31 * i1; l41 = n41 * i1
        d2 = n22 - l21 * n21; t32 = n32 - l31 * n21; t42 = n42 - l41 * n21
        i2 = f32(1) / d2
        l32 = t32 * i2; l42 = t42 * i2
        d3 = n33 - l31 * n31 - l32 * t32; t43 = n43 - l41 * n31 - l42 * t32
        i3 = f32(1) / d3
        l43 = t43 * i3
        d4 = n44 - l41 * n41 - l42 * t42 - l43 * t43
        # L^T x = e4: x4 = 1; x3 = -l43; x2 = -l42 - l32 x3; x1 = -l41 - l31 x3 - l21 x2
        w = np.ones(n, f32); z = -l43; y = -l42 - l32 * z; x = -l41 - l31 * z - l21 * y
        return x.astype(f32), y.astype(f32), z.astype(f32), w, (n11, d2, d3, d4)
    x, y, z, w, piv = nullvec(lam)
    for r in range(refine):
        # Rayleigh quotient
        kx = Kxx * x + Kxy * y + Kxz * z + Kxw * w; ky = Kxy * x + Kyy * y + Kyz * z + Kyw * w
        kz = Kxz * x + Kyz * y + Kzz * z + Kzw * w; kw = Kxw * x + Kyw * y + Kzw * z + Kww * w
        lam2 = ((x * kx + y * ky + z * kz + w * kw) / (x * x + y * y + z * z + w * w)).astype(f32)
        # the Rayleigh quotient is <= lambda_max: N would be indefinite by rounding; nudge up a few ulp
        lam = lam2
        x, y, z, w, piv = nullvec(lam)
    s = f32(2) / (x * x + y * y + z * z + w * w)
    xs, ys, zs = x * s, y * s, z * s
    R = np.empty((n, 3, 3), f32)
    R[:, 0, 0] = f32(1) - (y * ys + z * zs); R[:, 0, 1] = x * ys - w * zs; R[:, 0, 2] = x * zs + w * ys
    R[:, 1, 0] = x * ys + w * zs; R[:, 1, 1] = f32(1) - (x * xs + z * zs); R[:, 1, 2] = y * zs - w * xs
    R[:, 2, 0] = x * zs - w * ys; R[:, 2, 1] = y * zs + w * xs; R[:, 2, 2] = f32(1) - (x * xs + y * ys)
    R = R * sg[:, None, :]               # R = R' D
    if stats is not None:
        stats['piv'] = piv; stats['lam'] = lam; stats['w2'] = w * w * s / 2
    return R

if __name__ == '__main__':
    rng = np.random.default_rng(0)
    N = 400000
    M = rng.standard_normal((N, 3, 3)).astype(f32)
    Rref, S, d = ref_f64(M)
    gap = np.where(d < 0, S[:, 1] - S[:, 2], S[:, 1] + S[:, 2]) / S[:, 0]
    for iters in (3, 4, 5, 6, 8):
        for refine in (0, 1, 2):
            st = {}
            with np.errstate(all='ignore'):
                R = qpath(M, iters, refine, st)
            err = np.abs(R - Rref).reshape(N, -1).max(1)
            sc = err * gap
            bad = ~np.isfinite(err)
            print("newton %d refine %d: nan %d | err med %.2e p99 %.2e p99.9 %.2e max %.2e | scaled p50 %.2e p99 %.2e p99.9 %.2e max %.2e | rows scaled>3e-6: %d"
                  % (iters, refine, bad.sum(), np.nanmedian(err), np.nanquantile(err, .99), np.nanquantile(err, .999), np.nanmax(err),
                     np.nanmedian(sc), np.nanquantile(sc, .99), np.nanquantile(sc, .999), np.nanmax(sc), (sc > 3e-6).sum()))
