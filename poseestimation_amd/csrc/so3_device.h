// so3_device.h -- per-lane 3x3 "signed SVD" for gfx950, everything in VGPRs.
//
// What the reference computes (rotation_representation.py:199-205):
//     u, s, v = svd(m);  d = det(u v^T);  r = u diag(1,1,d) v^T
// What one lane computes here.  Write the SVD in its *signed* form
//     M = U' diag(s1, s2, s3') V^T,   U', V in SO(3),   s1, s2 >= |s3'|,   s3' = d * s3
// (U' = U diag(1,1,det U), V likewise, signs pushed into s3').  Then r = U' V^T with no explicit
// determinant flip, and because U' is a rotation its third column is u1 x u2.
//
//   1. prescale M by a power of two (exact) so squares neither overflow nor underflow;
//   2. kSweeps cyclic sweeps of one-sided (Hestenes) Jacobi on the columns of A = M:
//      A <- A J,  J a plane rotation that orthogonalises columns (p,q).  No V accumulation:
//      A = M V = U' S' holds implicitly because every J is orthogonal;
//   3. z = column of smallest norm (carries s3'), (x, y) = the other two in cyclic order;
//      u1 = x/|x|, u2 = Gram-Schmidt(y), u3 = u1 x u2;
//   4. v1 = M^T u1 / |.|, v2 = Gram-Schmidt(M^T u2), v3 = v1 x v2;   R = sum_k u_k v_k^T.
//
// One-sided Jacobi works on M itself (never forms M^T M), so small singular values keep their
// relative accuracy; measured against float64 LAPACK the result is closer than the reference's
// own float32 LAPACK path (tools/proto_jacobi.py, DESIGN.md section "accuracy").
// Rank <= 1 input (where the SVD is not unique) takes a rarely-executed divergent branch.
#pragma once
#include <hip/hip_runtime.h>

namespace so3 {

constexpr int kSweeps = 4;          // fixed; fp32 converges in 3 on Gaussian input (proto_jacobi.py)
constexpr float kDelta = 1e-18f;    // keeps the rotation well defined when alpha=beta, gamma=0
constexpr float kTinyNorm2 = 1e-30f;
constexpr float kTieBreak = 1.0f - 4e-6f;

struct V3 {
    float x, y, z;
};

__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ V3 scale(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 axpy(float s, V3 a, V3 b) {   // s*a + b
    return mk(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z));
}
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return mk(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
__device__ __forceinline__ V3 sel(bool c, V3 a, V3 b) { return mk(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }      // v_rsq_f32, 1 ulp
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }  // v_sqrt_f32, 1 ulp
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }    // v_rcp_f32, 1 ulp

// Orthogonalise columns p and q by a plane rotation (p, q) <- (c p + s q, c q - s p).
// With d = |p|^2 - |q|^2, g = 2 p.q, h = sqrt(d^2 + g^2):  (c, s) = (d + sgn(d) h, g) normalised,
// i.e. tan(theta) = g / (d + sgn(d) h), |theta| <= pi/4 (up to a common sign of both new columns,
// which a one-sided sweep does not care about).  Two transcendentals, no division.
__device__ __forceinline__ void rotate(V3 &p, V3 &q) {
    const float al = dot(p, p), be = dot(q, q), ga = dot(p, q);
    const float d = al - be;
    const float g = ga + ga;
    const float gg = g * g;
    const float h = fsqrt(fmaf(d, d, gg)) + kDelta;
    const float ae = d + copysignf(h, d);
    const float rw = rsq(fmaf(ae, ae, gg));
    const float c = ae * rw, s = g * rw;
    const V3 np = mk(fmaf(c, p.x, s * q.x), fmaf(c, p.y, s * q.y), fmaf(c, p.z, s * q.z));
    const V3 nq = mk(fmaf(c, q.x, -(s * p.x)), fmaf(c, q.y, -(s * p.y)), fmaf(c, q.z, -(s * p.z)));
    p = np;
    q = nq;
}

// A unit vector orthogonal to the unit vector u: e_k x u, k = index of the smallest |u_k| (z first).
__device__ __forceinline__ V3 any_perp(V3 u) {
    const float ax = fabsf(u.x), ay = fabsf(u.y), az = fabsf(u.z);
    V3 w;
    if (az <= ax && az <= ay) w = mk(-u.y, u.x, 0.f);
    else if (ay <= ax) w = mk(u.z, 0.f, -u.x);
    else w = mk(0.f, -u.z, u.y);
    return scale(w, rsq(dot(w, w)));
}

struct SignedSvd {
    V3 u1, u2, u3;      // columns of U' (right-handed)
    V3 v1, v2, v3;      // columns of V  (right-handed)
    float s1, s2, s3;   // s3 carries the sign; in units of the PRESCALED matrix
    float inv_scale;    // M_prescaled = M * 2^k ;  inv_scale = 2^k  (multiply gradients by it)
};

// m: row-major 3x3 (m[3*i+j]).  WANT_S: also fill s1,s2,s3 (backward needs them).
template <bool WANT_S>
__device__ __forceinline__ SignedSvd signed_svd(const float (&m_in)[9]) {
    SignedSvd o;
    // 1. exact power-of-two prescale: largest |entry| lands in [0.5, 1)
    float mx = fmaxf(fmaxf(fabsf(m_in[0]), fabsf(m_in[1])), fabsf(m_in[2]));
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(m_in[3]), fabsf(m_in[4])), fabsf(m_in[5])));
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(m_in[6]), fabsf(m_in[7])), fabsf(m_in[8])));
    const int ex = -__builtin_amdgcn_frexp_expf(mx);    // 0 for mx == 0; finite for inf/NaN too
    float m[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = ldexpf(m_in[i], ex);
    o.inv_scale = ldexpf(1.0f, ex);

    // 2. one-sided Jacobi on the columns
    V3 a0 = mk(m[0], m[3], m[6]), a1 = mk(m[1], m[4], m[7]), a2 = mk(m[2], m[5], m[8]);
#pragma unroll
    for (int sweep = 0; sweep < kSweeps; ++sweep) {
        rotate(a0, a1);
        rotate(a0, a2);
        rotate(a1, a2);
    }

    // 3. smallest column last, cyclic order kept (so det of the implied V stays +1)
    const float n0 = dot(a0, a0), n1 = dot(a1, a1), n2 = dot(a2, a2);
    // Ties (equal singular values to within a few ulp, e.g. diag(1,1,-1)) go to the LAST column, as
    // LAPACK's ordering does: the reference then maps a pure reflection to the identity.
    const float n2t = n2 * kTieBreak;
    const bool z2 = (n2t <= n0) && (n2t <= n1);
    const bool z0 = (n0 <= n1);
    const V3 x = sel(z2, a0, sel(z0, a1, a2));
    const V3 y = sel(z2, a1, sel(z0, a2, a0));
    const V3 z = sel(z2, a2, sel(z0, a0, a1));
    const float nx = z2 ? n0 : (z0 ? n1 : n2);

    V3 u1 = scale(x, rsq(nx));
    V3 w = axpy(-dot(u1, y), u1, y);
    float nw = dot(w, w);
    V3 u2 = scale(w, rsq(nw));
    const V3 mr0 = mk(m[0], m[1], m[2]), mr1 = mk(m[3], m[4], m[5]), mr2 = mk(m[6], m[7], m[8]);
    V3 t1 = axpy(u1.z, mr2, axpy(u1.y, mr1, scale(mr0, u1.x)));     // M^T u1 = s1 v1
    V3 t2 = axpy(u2.z, mr2, axpy(u2.y, mr1, scale(mr0, u2.x)));     // M^T u2 = s2 v2
    float nt1 = dot(t1, t1);
    V3 v1 = scale(t1, rsq(nt1));
    V3 r2 = axpy(-dot(v1, t2), v1, t2);
    float nr2 = dot(r2, r2);
    V3 v2 = scale(r2, rsq(nr2));

    // Rank <= 1 (or all-zero) input: the frame is not unique; pick one deterministically.
    // (`<=` comparisons are false for NaN, so NaN input flows through the fast path to NaN output.)
    if (__builtin_expect(nx <= kTinyNorm2 || nw <= kTinyNorm2 || nt1 <= kTinyNorm2 || nr2 <= kTinyNorm2, 0)) {
        const bool b0 = (n0 >= n1) && (n0 >= n2);
        const bool b1 = (n1 >= n2);
        const V3 big = sel(b0, a0, sel(b1, a1, a2));
        const float nb = b0 ? n0 : (b1 ? n1 : n2);
        if (nb <= kTinyNorm2) {                      // M == 0  ->  identity (matches the reference)
            u1 = mk(1.f, 0.f, 0.f);
            v1 = u1;
        } else {
            u1 = scale(big, rsq(nb));
            t1 = axpy(u1.z, mr2, axpy(u1.y, mr1, scale(mr0, u1.x)));
            v1 = scale(t1, rsq(dot(t1, t1)));
        }
        u2 = any_perp(u1);
        v2 = any_perp(v1);
        if (WANT_S) { nt1 = nb; nr2 = 0.f; }
    }
    o.u1 = u1; o.u2 = u2; o.u3 = cross(u1, u2);
    o.v1 = v1; o.v2 = v2; o.v3 = cross(v1, v2);
    if (WANT_S) {
        // s_k = u_k^T M v_k; cheaper: |M^T u_k| for k = 1,2 and u3 . z for the signed one.
        o.s1 = nt1 * rsq(fmaxf(nt1, kTinyNorm2));
        o.s2 = nr2 * rsq(fmaxf(nr2, kTinyNorm2));
        o.s3 = dot(o.u3, z);
    } else {
        o.s1 = o.s2 = o.s3 = 0.f;
    }
    return o;
}

// R = U' V^T, row-major.
__device__ __forceinline__ void rotation_from(const SignedSvd &f, float (&r)[9]) {
    const float ux[3] = {f.u1.x, f.u2.x, f.u3.x}, uy[3] = {f.u1.y, f.u2.y, f.u3.y}, uz[3] = {f.u1.z, f.u2.z, f.u3.z};
    const V3 v[3] = {f.v1, f.v2, f.v3};
    V3 r0 = scale(v[0], ux[0]), r1 = scale(v[0], uy[0]), r2 = scale(v[0], uz[0]);
#pragma unroll
    for (int k = 1; k < 3; ++k) {
        r0 = axpy(ux[k], v[k], r0);
        r1 = axpy(uy[k], v[k], r1);
        r2 = axpy(uz[k], v[k], r2);
    }
    r[0] = r0.x; r[1] = r0.y; r[2] = r0.z;
    r[3] = r1.x; r[4] = r1.y; r[5] = r1.z;
    r[6] = r2.x; r[7] = r2.y; r[8] = r2.z;
}

// sign(det M) evaluated in float64 (products of floats are exact in double) -> flip flag.
__device__ __forceinline__ bool det_negative(const float (&m)[9]) {
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    return det < 0.0;
}

// dM = U' Bm V^T for upstream G (row-major), Bm_ij = (A_ij - A_ji)/(s_i + s_j), A = U'^T G V.
__device__ __forceinline__ void project_backward(const SignedSvd &f, const float (&g)[9], float (&dm)[9]) {
    const V3 g0 = mk(g[0], g[1], g[2]), g1 = mk(g[3], g[4], g[5]), g2 = mk(g[6], g[7], g[8]);
    // G v_j
    const V3 gv1 = mk(dot(g0, f.v1), dot(g1, f.v1), dot(g2, f.v1));
    const V3 gv2 = mk(dot(g0, f.v2), dot(g1, f.v2), dot(g2, f.v2));
    const V3 gv3 = mk(dot(g0, f.v3), dot(g1, f.v3), dot(g2, f.v3));
    const float a12 = dot(f.u1, gv2), a21 = dot(f.u2, gv1);
    const float a13 = dot(f.u1, gv3), a31 = dot(f.u3, gv1);
    const float a23 = dot(f.u2, gv3), a32 = dot(f.u3, gv2);
    const float floor_ = fmaf(1e-12f, f.s1, 1e-30f);
    const float k = f.inv_scale;                     // singular values are in prescaled units
    const float b12 = (a12 - a21) * k * frcp(fmaxf(f.s1 + f.s2, floor_));
    const float b13 = (a13 - a31) * k * frcp(fmaxf(f.s1 + f.s3, floor_));
    const float b23 = (a23 - a32) * k * frcp(fmaxf(f.s2 + f.s3, floor_));
    // T = U' Bm : t1 = -b12 u2 - b13 u3 ; t2 = b12 u1 - b23 u3 ; t3 = b13 u1 + b23 u2
    const V3 t1 = axpy(-b12, f.u2, scale(f.u3, -b13));
    const V3 t2 = axpy(b12, f.u1, scale(f.u3, -b23));
    const V3 t3 = axpy(b13, f.u1, scale(f.u2, b23));
    const V3 r0 = axpy(t3.x, f.v3, axpy(t2.x, f.v2, scale(f.v1, t1.x)));
    const V3 r1 = axpy(t3.y, f.v3, axpy(t2.y, f.v2, scale(f.v1, t1.y)));
    const V3 r2 = axpy(t3.z, f.v3, axpy(t2.z, f.v2, scale(f.v1, t1.z)));
    dm[0] = r0.x; dm[1] = r0.y; dm[2] = r0.z;
    dm[3] = r1.x; dm[4] = r1.y; dm[5] = r1.z;
    dm[6] = r2.x; dm[7] = r2.y; dm[8] = r2.z;
}

}  // namespace so3
