"""Prototype 2 (numpy float32): SO(3) projection = dominant eigenvector of Davenport's 4x4 matrix K(M).
lambda_max by Laguerre iteration on the characteristic quartic, eigenvector = largest column of adj(lambda I - K),
one Rayleigh-quotient refinement.  Rows the method cannot do accurately ("hard") are flagged for the Jacobi path."""
import numpy as np, sys
f32 = np.float32

def ref_f64(M):
    M = M.astype(np.float64)
    U, S, Vt = np.linalg.svd(M)
    d = np.sign(np.linalg.det(U @ Vt)); d[d == 0] = 1
    D = np.zeros_like(M); D[:, 0, 0] = 1; D[:, 1, 1] = 1; D[:, 2, 2] = d
    return U @ D @ Vt, S, d

def adj_sym4(n):
    """adjugate of symmetric 4x4 given as dict of 10 arrays n[(i,j)], i<=j.  Returns dict of 10."""
    g = lambda i, j: n[(i, j)] if i <= j else n[(j, i)]
    # 2x2 minors of rows (0,1): s, rows (2,3): c  (standard 4x4 inverse by complementary minors)
    a = [[g(i, j) for j in range(4)] for i in range(4)]
    s0 = a[0][0]*a[1][1]-a[1][0]*a[0][1]; s1 = a[0][0]*a[1][2]-a[1][0]*a[0][2]; s2 = a[0][0]*a[1][3]-a[1][0]*a[0][3]
    s3 = a[0][1]*a[1][2]-a[1][1]*a[0][2]; s4 = a[0][1]*a[1][3]-a[1][1]*a[0][3]; s5 = a[0][2]*a[1][3]-a[1][2]*a[0][3]
    c5 = a[2][2]*a[3][3]-a[3][2]*a[2][3]; c4 = a[2][1]*a[3][3]-a[3][1]*a[2][3]; c3 = a[2][1]*a[3][2]-a[3][1]*a[2][2]
    c2 = a[2][0]*a[3][3]-a[3][0]*a[2][3]; c1 = a[2][0]*a[3][2]-a[3][0]*a[2][2]; c0 = a[2][0]*a[3][1]-a[3][0]*a[2][1]
    b = {}
    b[(0,0)] =  a[1][1]*c5 - a[1][2]*c4 + a[1][3]*c3
    b[(0,1)] = -a[0][1]*c5 + a[0][2]*c4 - a[0][3]*c3
    b[(0,2)] =  a[3][1]*s5 - a[3][2]*s4 + a[3][3]*s3
    b[(0,3)] = -a[2][1]*s5 + a[2][2]*s4 - a[2][3]*s3
    b[(1,1)] =  a[0][0]*c5 - a[0][2]*c2 + a[0][3]*c1
    b[(1,2)] = -a[3][0]*s5 + a[3][2]*s2 - a[3][3]*s1
    b[(1,3)] =  a[2][0]*s5 - a[2][2]*s2 + a[2][3]*s1
    b[(2,2)] =  a[3][0]*s4 - a[3][1]*s2 + a[3][3]*s0
    b[(2,3)] = -a[2][0]*s4 + a[2][1]*s2 - a[2][3]*s0
    b[(3,3)] =  a[2][0]*s3 - a[2][1]*s1 + a[2][2]*s0
    det = s0*c5 - s1*c4 + s2*c3 + s3*c2 - s4*c1 + s5*c0
    return {k: v.astype(f32) for k, v in b.items()}, det.astype(f32)

def qpath(M, lag_iters=4, refine=1, hard_tau=1e-3, stats=None, conv=3e-4):
    M = M.astype(f32); n = len(M)
    mx = np.abs(M).reshape(n, -1).max(1)
    e = np.where(mx > 0, np.floor(np.log2(np.maximum(mx, 1e-45))) + 1, 0)
    M = (M * (f32(2.0) ** (-e)).astype(f32)[:, None, None]).astype(f32)
    m = lambda i, j: M[:, i, j]
    K = {(0,0): m(0,0)+m(1,1)+m(2,2), (1,1): m(0,0)-m(1,1)-m(2,2), (2,2): -m(0,0)+m(1,1)-m(2,2), (3,3): -m(0,0)-m(1,1)+m(2,2),
         (0,1): m(2,1)-m(1,2), (0,2): m(0,2)-m(2,0), (0,3): m(1,0)-m(0,1), (1,2): m(0,1)+m(1,0), (1,3): m(0,2)+m(2,0), (2,3): m(1,2)+m(2,1)}
    # coefficients
    f = (M*M).reshape(n, -1).sum(1, dtype=f32)
    C = np.empty_like(M)
    C[:,0,0] = m(1,1)*m(2,2)-m(1,2)*m(2,1); C[:,0,1] = m(1,2)*m(2,0)-m(1,0)*m(2,2); C[:,0,2] = m(1,0)*m(2,1)-m(1,1)*m(2,0)
    C[:,1,0] = m(0,2)*m(2,1)-m(0,1)*m(2,2); C[:,1,1] = m(0,0)*m(2,2)-m(0,2)*m(2,0); C[:,1,2] = m(0,1)*m(2,0)-m(0,0)*m(2,1)
    C[:,2,0] = m(0,1)*m(1,2)-m(0,2)*m(1,1); C[:,2,1] = m(0,2)*m(1,0)-m(0,0)*m(1,2); C[:,2,2] = m(0,0)*m(1,1)-m(0,1)*m(1,0)
    det = (m(0,0)*C[:,0,0] + m(0,1)*C[:,0,1] + m(0,2)*C[:,0,2]).astype(f32)
    cf = (C*C).reshape(n, -1).sum(1, dtype=f32)
    c2 = f32(-2)*f; c1 = f32(-8)*det; c0 = (f*f - f32(4)*cf).astype(f32)
    lam = np.sqrt(f32(3)*f).astype(f32)
    with np.errstate(all='ignore'):
        for it in range(lag_iters):
            l2 = lam*lam
            P = ((l2 + c2)*lam + c1)*lam + c0
            dP = (f32(4)*l2 + f32(2)*c2)*lam + c1
            ddP = f32(12)*l2 + f32(2)*c2
            H = f32(9)*dP*dP - f32(12)*P*ddP           # (n-1)((n-1)P'^2 - n P P''), n = 4  -> 3(3P'^2-4PP'')
            den = dP + np.sqrt(np.maximum(H, f32(0)))
            lam = (lam - f32(4)*P/den).astype(f32)
        def eigvec(lam):
            N = {k: (-v).astype(f32) for k, v in K.items()}
            for i in range(4): N[(i,i)] = (lam - K[(i,i)]).astype(f32)
            A, dN = adj_sym4(N)
            diag = np.stack([A[(0,0)], A[(1,1)], A[(2,2)], A[(3,3)]], 1)
            j = diag.argmax(1)
            g = lambda i, k: A[(i,k)] if i <= k else A[(k,i)]
            q = np.stack([np.choose(j, [g(i,0), g(i,1), g(i,2), g(i,3)]) for i in range(4)], 1).astype(f32)
            tr = diag.sum(1, dtype=f32)
            return q, tr, diag.max(1)
        q, tr, dmax = eigvec(lam)
        lam_first = lam
        for r in range(refine):
            kq = np.stack([sum(( (K[(i,k)] if i<=k else K[(k,i)]) * q[:,k] for k in range(4))) for i in range(4)], 1).astype(f32)
            lam = ((q*kq).sum(1, dtype=f32) / (q*q).sum(1, dtype=f32)).astype(f32)
            q, tr, dmax = eigvec(lam)
        w, x, y, z = q[:,0], q[:,1], q[:,2], q[:,3]
        s = f32(2)/(q*q).sum(1, dtype=f32)
        xs, ys, zs = x*s, y*s, z*s
        R = np.empty((n,3,3), f32)
        R[:,0,0] = f32(1)-(y*ys+z*zs); R[:,0,1] = x*ys-w*zs; R[:,0,2] = x*zs+w*ys
        R[:,1,0] = x*ys+w*zs; R[:,1,1] = f32(1)-(x*xs+z*zs); R[:,1,2] = y*zs-w*xs
        R[:,2,0] = x*zs-w*ys; R[:,2,1] = y*zs+w*xs; R[:,2,2] = f32(1)-(x*xs+y*ys)
        # hard rows: product of the three gaps (= trace of the adjugate) small against lambda^3, or not finite
        # hard rows: (a) product of the three gaps (= trace of the adjugate) small against lambda^3;
        # (b) the Rayleigh quotient moved lambda by more than conv * (lower bound of the gap): Laguerre had not converged;
        # (c) not finite
        l3 = lam*lam*lam
        hard = ~(tr > f32(hard_tau)*l3) | ~(np.abs(lam_first - lam)*f32(4)*lam*lam <= f32(conv)*tr) | ~np.isfinite(R).all((1,2))
    if stats is not None: stats.update(lam=lam, tr=tr, f=f)
    return R, hard


def families(n, rng):
    def rot(k):
        return ref_f64(rng.standard_normal((k,3,3)))[0]
    def dm(d): 
        D = np.zeros((n,3,3)); D[:,0,0]=d[:,0]; D[:,1,1]=d[:,1]; D[:,2,2]=d[:,2]; return D
    yield "gaussian", rng.standard_normal((n,3,3))
    for e in (1e-1,1e-3,1e-5,1e-7):
        d = np.ones((n,3)); d[:,1] = 1-e*rng.random(n); d[:,2] = 1-2*e*rng.random(n)
        yield "clustered singular values, spread %.0e"%e, rot(n)@dm(d)@rot(n)
    for e in (1e-2,1e-4,1e-6):
        d = np.ones((n,3)); d[:,1]=e; d[:,2]=e*e
        yield "graded 1, %.0e, %.0e"%(e,e*e), rot(n)@dm(d)@rot(n)
    for e in (1e-1,1e-3,1e-5):
        yield "rotation + %.0e noise"%e, rot(n)+e*rng.standard_normal((n,3,3))
    a = rng.standard_normal((n,3,3))
    yield "symmetric", a+a.transpose(0,2,1)
    yield "antisymmetric + 1e-3 I", a-a.transpose(0,2,1)+1e-3*np.eye(3)
    yield "small integers", rng.integers(-3,4,(n,3,3)).astype(np.float64)
    u, v = rng.standard_normal((n,3,1)).astype(f32), rng.standard_normal((n,1,3)).astype(f32)
    yield "outer products", (u@v).astype(np.float64)
    yield "outer products of small integers", (rng.integers(-3,4,(n,3,1)).astype(f32)@rng.integers(-3,4,(n,1,3)).astype(f32)).astype(np.float64)
    yield "nine equal entries", np.broadcast_to(rng.standard_normal((n,1,1)),(n,3,3)).copy()
    yield "rank two", np.concatenate((a[:,:2], a[:,:1]+a[:,1:2]),1)
    yield "scaled 1e+18", 1e18*rng.standard_normal((n,3,3))
    yield "scaled 1e-18", 1e-18*rng.standard_normal((n,3,3))
    yield "reflected gaussian (det<0)", a*np.where(np.linalg.det(a)>0,-1,1)[:,None,None]
    yield "near rotations (0.02 noise)", rot(n)+0.02*rng.standard_normal((n,3,3))

if __name__ == '__main__':
    rng = np.random.default_rng(1)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    it = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    tau = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
    conv = float(sys.argv[4]) if len(sys.argv) > 4 else 3e-4
    for name, M64 in families(n, rng):
        M = M64.astype(f32)
        Rref, S, d = ref_f64(M)
        gap = np.where(d < 0, S[:,1]-S[:,2], S[:,1]+S[:,2])/np.maximum(S[:,0],1e-300)
        R, hard = qpath(M, it, 1, tau, conv=conv)
        ok = ~hard
        if ok.any():
            err = np.abs(R-Rref).reshape(n,-1).max(1); sc = err*gap
            orth = np.abs(np.einsum('nij,nik->njk', R[ok].astype(np.float64), R[ok].astype(np.float64)) - np.eye(3)).reshape(ok.sum(), -1).max(1)
            print("%-40s hard %.2e | easy: err med %.2e max %.2e | scaled p99.9 %.2e max %.2e | >3e-6: %d | orth %.1e | min gap easy %.1e"
                  % (name, hard.mean(), np.median(err[ok]), err[ok].max(), np.quantile(sc[ok], .999), sc[ok].max(), (sc[ok] > 3e-6).sum(), orth.max(), gap[ok].min()))
        else:
            print("%-40s hard 100%%" % name)
