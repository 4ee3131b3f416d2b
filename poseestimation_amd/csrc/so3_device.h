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
// own float32 LAPACK path (docs/history/tools.tar.gz:tools/proto_jacobi.py, DESIGN.md section "accuracy").
// Rank <= 1 input (where the SVD is not unique) takes a rarely-executed divergent branch.
//
// The arithmetic is written once, generic over the "scalar" type T:
//   T = float  : one matrix per lane;
//   T = f32x2  : TWO matrices per lane, one in each half of a 64-bit register pair.  A Jacobi sweep
//                is one long dependent chain, and a gfx950 SIMD issues dependent VALU instructions of
//                one wave at only ~0.5/ns against ~0.9/ns for independent ones
//                (tools/ubench/valu_rates.hip).  Packing two independent matrices into v_pk_fma_f32 /
//                v_pk_mul_f32 / v_pk_add_f32 (which do two lanes' worth of work per issue slot) restores
//                the full rate without relying on the scheduler to interleave two scalar streams.
#pragma once
// SO3_HOST_MODEL: the same templates compiled for the host (oracle/kernel_model.cpp: the CPU test suite runs the
// kernel's own algorithm on adversarial input without a GPU).  Only the hardware instructions below differ:
// libm stands in for v_rsq_f32 / v_sqrt_f32 / v_rcp_f32 (correctly rounded instead of 1 ulp) and a "wave" is one lane.
#ifdef SO3_HOST_MODEL
#include <algorithm>
#include <cmath>
#define __device__
#define __forceinline__ inline
#else
#include <hip/hip_runtime.h>
#endif

namespace so3 {

namespace hw {
#ifdef SO3_HOST_MODEL
using std::min;
inline float rsq(float x) { return 1.0f / std::sqrt(x); }
inline float sqrt(float x) { return std::sqrt(x); }
inline float rcp(float x) { return 1.0f / x; }
inline float cos_rev(float x) { return std::cos(6.28318530717958647692f * x); }
inline int frexp_exp(float x) { int e = 0; if (x != 0.0f && std::isfinite(x)) std::frexp(x, &e); return e; }
inline int frexp_exp(double x) { int e = 0; if (x != 0.0 && std::isfinite(x)) std::frexp(x, &e); return e; }
inline bool any_lane(bool p) { return p; }
inline float med3(float a, float b, float c) { return std::max(std::min(a, b), std::min(std::max(a, b), c)); }
#else
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }       // v_rsq_f32, 1 ulp
__device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }     // v_sqrt_f32, 1 ulp
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }       // v_rcp_f32, 1 ulp
__device__ __forceinline__ float cos_rev(float x) { return __builtin_amdgcn_cosf(x); }   // v_cos_f32: cos(2 pi x), |x| <= 256
__device__ __forceinline__ int frexp_exp(float x) { return __builtin_amdgcn_frexp_expf(x); }   // 0 for 0, inf and NaN
__device__ __forceinline__ int frexp_exp(double x) { return __builtin_amdgcn_frexp_exp(x); }
__device__ __forceinline__ bool any_lane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
__device__ __forceinline__ float med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }   // v_med3_f32
#endif
}  // namespace hw

// Sweep schedule: kSweeps fixed cyclic sweeps, then -- if any matrix held by the wave still has a relative
// off-orthogonality of its (0,1) pair above kResidualTol (a wave-uniform branch) -- one more rotation of that pair,
// kept by the matrices that failed.  On Gaussian input 99.93 % of the rows are below 1e-5 after three sweeps
// (docs/history/tools.tar.gz:tools/proto_jacobi.py), so about one wave round in eight takes the branch.
constexpr int kSweeps = 3;
constexpr float kResidualTol2 = 0.5e-10f;  // (0.7e-5)^2 on  gamma_01^2 / (|a_0|^2 |a_1|^2)
constexpr float kDelta = 1e-18f;    // keeps the rotation well defined when alpha=beta, gamma=0
constexpr float kTinyNorm2 = 1e-30f;
constexpr float kTieBreak = 1.0f - 4e-6f;
// The same four for float64 arithmetic (so3_project_*_f64): tolerance (1e-14)^2, and sweeps run until it is met.
template <class S> struct Consts {          // S = float
    static constexpr float tol2 = kResidualTol2, delta = kDelta, tiny = kTinyNorm2, tie = kTieBreak, bwd_rel = 1e-12f, bwd_abs = 1e-30f;
};
template <> struct Consts<double> {
    static constexpr double tol2 = 1e-28, delta = 1e-150, tiny = 1e-280, tie = 1.0 - 1e-13, bwd_rel = 1e-24, bwd_abs = 1e-290;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
// Per-half predicate of a packed pair.
#ifdef SO3_HOST_MODEL
struct bool2 {
    bool x, y;
};
__device__ __forceinline__ bool2 operator&(bool2 a, bool2 b) { return bool2{a.x && b.x, a.y && b.y}; }
__device__ __forceinline__ bool2 operator|(bool2 a, bool2 b) { return bool2{a.x || b.x, a.y || b.y}; }
__device__ __forceinline__ bool2 operator^(bool2 a, bool2 b) { return bool2{a.x != b.x, a.y != b.y}; }
#else
// On the device a predicate IS its wave's lane mask, one 64-bit word per half (an SGPR pair): v_cmp writes it there, and / or /
// not / "any lane" run on the scalar unit, and a select takes it back as v_cndmask's condition (inverse ballot).  Round 4 kept two
// bools; wherever a predicate crossed a branch or a return the compiler packed the pair into a 16-bit VGPR (v_cndmask 0/1, v_or,
// v_cmp_ne_u16: four vector instructions per "does any row ...", thirty per round of K1).  Every use sits in wave-uniform control
// flow (the engine's loop), which the ballots need.
struct bool2 {
    unsigned long long x, y;
};
__device__ __forceinline__ bool2 operator&(bool2 a, bool2 b) { return bool2{a.x & b.x, a.y & b.y}; }
__device__ __forceinline__ bool2 operator|(bool2 a, bool2 b) { return bool2{a.x | b.x, a.y | b.y}; }
__device__ __forceinline__ bool2 operator^(bool2 a, bool2 b) { return bool2{a.x ^ b.x, a.y ^ b.y}; }
#endif

// ---- scalar-type traits ------------------------------------------------------------------------------
template <class T> struct Tr;
template <> struct Tr<float> {
    typedef bool mask;
    typedef int ivec;
    typedef float scalar;
    static constexpr int kLanes = 1;
    static __device__ __forceinline__ float splat(float v) { return v; }
    static __device__ __forceinline__ float get(float v, int) { return v; }
    static __device__ __forceinline__ void set(float &v, int, float x) { v = x; }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return fmaf(a, b, c); }
    static __device__ __forceinline__ float abs(float a) { return fabsf(a); }
    static __device__ __forceinline__ float max(float a, float b) { return fmaxf(a, b); }
    static __device__ __forceinline__ float vmin(float a, float b) { return fminf(a, b); }
    // clamp / max as ONE v_med3_f32 (fmaxf / fminf each come with a canonicalising v_max x, x in front when x is not known to be quiet);
    // NaN in: some finite bound out -- only used where a NaN row is declared hard anyway
    static __device__ __forceinline__ float clamp(float x, float lo, float hi) { return hw::med3(x, lo, hi); }
    static __device__ __forceinline__ float max_fast(float a, float b) { return hw::med3(a, b, __builtin_inff()); }
    static __device__ __forceinline__ float cos_rev(float x) { return hw::cos_rev(x); }          // cos(2 pi x), hardware accuracy
    static __device__ __forceinline__ float copysign(float a, float b) { return copysignf(a, b); }
    static __device__ __forceinline__ float rsq(float x) { return hw::rsq(x); }
    static __device__ __forceinline__ float sqrt(float x) { return hw::sqrt(x); }
    static __device__ __forceinline__ float rcp(float x) { return hw::rcp(x); }
    static __device__ __forceinline__ float sin(float x) { return sinf(x); }                     // full-range libm forms
    static __device__ __forceinline__ float cos(float x) { return cosf(x); }
    static __device__ __forceinline__ void sincos(float x, float &s, float &c) { sincosf(x, &s, &c); }   // one range reduction
    // exponent that brings x into [0.5, 1), clamped so that 2^e stays finite (denormal input)
    static __device__ __forceinline__ int neg_frexp_exp(float x) { using namespace hw; return min(-hw::frexp_exp(x), 126); }
    static __device__ __forceinline__ float ldexp(float x, int e) { return ldexpf(x, e); }
    static __device__ __forceinline__ float sel(bool c, float a, float b) { return c ? a : b; }
    static __device__ __forceinline__ bool le(float a, float b) { return a <= b; }
    static __device__ __forceinline__ bool ge(float a, float b) { return a >= b; }
    static __device__ __forceinline__ bool gt(float a, float b) { return a > b; }
    static __device__ __forceinline__ bool any(bool m) { return m; }
    static __device__ __forceinline__ bool wave_any(bool m) { return hw::any_lane(m); }     // wave-uniform: on any lane of the wave
    static __device__ __forceinline__ bool lane_of(bool m, int) { return m; }
    static __device__ __forceinline__ bool mnot(bool m) { return !m; }
};
template <> struct Tr<f32x2> {
    typedef bool2 mask;
    typedef i32x2 ivec;
    typedef float scalar;
    static constexpr int kLanes = 2;
    static __device__ __forceinline__ f32x2 splat(float v) { return f32x2{v, v}; }
    static __device__ __forceinline__ float get(f32x2 v, int i) { return i ? v.y : v.x; }
    static __device__ __forceinline__ void set(f32x2 &v, int i, float x) { if (i) v.y = x; else v.x = x; }
    static __device__ __forceinline__ f32x2 fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
    static __device__ __forceinline__ f32x2 abs(f32x2 a) { return __builtin_elementwise_abs(a); }
    static __device__ __forceinline__ f32x2 max(f32x2 a, f32x2 b) { return f32x2{fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
    static __device__ __forceinline__ f32x2 vmin(f32x2 a, f32x2 b) { return f32x2{fminf(a.x, b.x), fminf(a.y, b.y)}; }
    static __device__ __forceinline__ f32x2 clamp(f32x2 x, f32x2 lo, f32x2 hi) { return f32x2{hw::med3(x.x, lo.x, hi.x), hw::med3(x.y, lo.y, hi.y)}; }
    static __device__ __forceinline__ f32x2 max_fast(f32x2 a, f32x2 b) { return f32x2{hw::med3(a.x, b.x, __builtin_inff()), hw::med3(a.y, b.y, __builtin_inff())}; }
    static __device__ __forceinline__ f32x2 cos_rev(f32x2 x) { return f32x2{hw::cos_rev(x.x), hw::cos_rev(x.y)}; }
    static __device__ __forceinline__ f32x2 copysign(f32x2 a, f32x2 b) { return f32x2{copysignf(a.x, b.x), copysignf(a.y, b.y)}; }
    static __device__ __forceinline__ f32x2 rsq(f32x2 x) { return f32x2{hw::rsq(x.x), hw::rsq(x.y)}; }
    static __device__ __forceinline__ f32x2 sqrt(f32x2 x) { return f32x2{hw::sqrt(x.x), hw::sqrt(x.y)}; }
    static __device__ __forceinline__ f32x2 rcp(f32x2 x) { return f32x2{hw::rcp(x.x), hw::rcp(x.y)}; }
    static __device__ __forceinline__ f32x2 sin(f32x2 x) { return f32x2{sinf(x.x), sinf(x.y)}; }
    static __device__ __forceinline__ f32x2 cos(f32x2 x) { return f32x2{cosf(x.x), cosf(x.y)}; }
    static __device__ __forceinline__ void sincos(f32x2 x, f32x2 &s, f32x2 &c) {
        float s0, c0, s1, c1;
        sincosf(x.x, &s0, &c0);
        sincosf(x.y, &s1, &c1);
        s = f32x2{s0, s1};
        c = f32x2{c0, c1};
    }
    static __device__ __forceinline__ i32x2 neg_frexp_exp(f32x2 x) {
        using namespace hw;
        return i32x2{min(-hw::frexp_exp(x.x), 126), min(-hw::frexp_exp(x.y), 126)};
    }
    static __device__ __forceinline__ f32x2 ldexp(f32x2 x, i32x2 e) { return f32x2{ldexpf(x.x, e.x), ldexpf(x.y, e.y)}; }
#ifdef SO3_HOST_MODEL
    static __device__ __forceinline__ f32x2 sel(bool2 c, f32x2 a, f32x2 b) { return f32x2{c.x ? a.x : b.x, c.y ? a.y : b.y}; }
    static __device__ __forceinline__ bool2 le(f32x2 a, f32x2 b) { return bool2{a.x <= b.x, a.y <= b.y}; }
    static __device__ __forceinline__ bool2 ge(f32x2 a, f32x2 b) { return bool2{a.x >= b.x, a.y >= b.y}; }
    static __device__ __forceinline__ bool2 gt(f32x2 a, f32x2 b) { return bool2{a.x > b.x, a.y > b.y}; }
    static __device__ __forceinline__ bool any(bool2 m) { return m.x || m.y; }
    static __device__ __forceinline__ bool wave_any(bool2 m) { return m.x || m.y; }
    static __device__ __forceinline__ bool lane_of(bool2 m, int i) { return i ? m.y : m.x; }
    static __device__ __forceinline__ bool2 mnot(bool2 m) { return bool2{!m.x, !m.y}; }
#else
    static __device__ __forceinline__ bool mine(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
    static __device__ __forceinline__ f32x2 sel(bool2 c, f32x2 a, f32x2 b) { return f32x2{mine(c.x) ? a.x : b.x, mine(c.y) ? a.y : b.y}; }
    static __device__ __forceinline__ bool2 le(f32x2 a, f32x2 b) { return bool2{__builtin_amdgcn_ballot_w64(a.x <= b.x), __builtin_amdgcn_ballot_w64(a.y <= b.y)}; }
    static __device__ __forceinline__ bool2 ge(f32x2 a, f32x2 b) { return bool2{__builtin_amdgcn_ballot_w64(a.x >= b.x), __builtin_amdgcn_ballot_w64(a.y >= b.y)}; }
    static __device__ __forceinline__ bool2 gt(f32x2 a, f32x2 b) { return bool2{__builtin_amdgcn_ballot_w64(a.x > b.x), __builtin_amdgcn_ballot_w64(a.y > b.y)}; }
    // (a complement sets the bits of lanes that are switched off as well: the live ones are what counts)
    static __device__ __forceinline__ bool any(bool2 m) { return ((m.x | m.y) & __builtin_amdgcn_ballot_w64(true)) != 0; }
    static __device__ __forceinline__ bool wave_any(bool2 m) { return any(m); }
    static __device__ __forceinline__ bool lane_of(bool2 m, int i) { return mine(i ? m.y : m.x); }
    static __device__ __forceinline__ bool2 mnot(bool2 m) { return bool2{~m.x, ~m.y}; }
#endif
};

template <> struct Tr<double> {           // one matrix per lane in float64 (so3_project_*_f64; not a benchmark path)
    typedef bool mask;
    typedef int ivec;
    typedef double scalar;
    static constexpr int kLanes = 1;
    static __device__ __forceinline__ double splat(double v) { return v; }
    static __device__ __forceinline__ double get(double v, int) { return v; }
    static __device__ __forceinline__ void set(double &v, int, double x) { v = x; }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    static __device__ __forceinline__ double abs(double a) { return __builtin_fabs(a); }
    static __device__ __forceinline__ double max(double a, double b) { return __builtin_fmax(a, b); }
    static __device__ __forceinline__ double vmin(double a, double b) { return __builtin_fmin(a, b); }
    static __device__ __forceinline__ double clamp(double x, double lo, double hi) { return __builtin_fmax(__builtin_fmin(x, hi), lo); }
    static __device__ __forceinline__ double max_fast(double a, double b) { return __builtin_fmax(a, b); }
    static __device__ __forceinline__ double cos_rev(double x) { return ::cos(6.28318530717958647692 * x); }
    static __device__ __forceinline__ double copysign(double a, double b) { return __builtin_copysign(a, b); }
    static __device__ __forceinline__ double sqrt(double x) { return __builtin_sqrt(x); }          // correctly rounded
    static __device__ __forceinline__ double rsq(double x) { return 1.0 / __builtin_sqrt(x); }
    static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
    static __device__ __forceinline__ int neg_frexp_exp(double x) { using namespace hw; return min(-hw::frexp_exp(x), 1022); }
    static __device__ __forceinline__ double ldexp(double x, int e) { return ::ldexp(x, e); }
    static __device__ __forceinline__ double sel(bool c, double a, double b) { return c ? a : b; }
    static __device__ __forceinline__ bool le(double a, double b) { return a <= b; }
    static __device__ __forceinline__ bool ge(double a, double b) { return a >= b; }
    static __device__ __forceinline__ bool gt(double a, double b) { return a > b; }
    static __device__ __forceinline__ bool any(bool m) { return m; }
    static __device__ __forceinline__ bool wave_any(bool m) { return hw::any_lane(m); }
    static __device__ __forceinline__ bool lane_of(bool m, int) { return m; }
    static __device__ __forceinline__ bool mnot(bool m) { return !m; }
};

// ---- 3-vectors over T -----------------------------------------------------------------------------
template <class T> struct V3 {
    T x, y, z;
};
template <class T> __device__ __forceinline__ V3<T> mk(T x, T y, T z) { return V3<T>{x, y, z}; }
template <class T> __device__ __forceinline__ T dot(V3<T> a, V3<T> b) {
    return Tr<T>::fma(a.z, b.z, Tr<T>::fma(a.y, b.y, a.x * b.x));
}
template <class T> __device__ __forceinline__ V3<T> scale(V3<T> a, T s) { return mk<T>(a.x * s, a.y * s, a.z * s); }
template <class T> __device__ __forceinline__ V3<T> axpy(T s, V3<T> a, V3<T> b) {   // s*a + b
    return mk<T>(Tr<T>::fma(s, a.x, b.x), Tr<T>::fma(s, a.y, b.y), Tr<T>::fma(s, a.z, b.z));
}
template <class T> __device__ __forceinline__ V3<T> cross(V3<T> a, V3<T> b) {
    return mk<T>(Tr<T>::fma(a.y, b.z, -(a.z * b.y)), Tr<T>::fma(a.z, b.x, -(a.x * b.z)), Tr<T>::fma(a.x, b.y, -(a.y * b.x)));
}
template <class T> __device__ __forceinline__ V3<T> sel(typename Tr<T>::mask c, V3<T> a, V3<T> b) {
    return mk<T>(Tr<T>::sel(c, a.x, b.x), Tr<T>::sel(c, a.y, b.y), Tr<T>::sel(c, a.z, b.z));
}

// ---- the Jacobi path (section 3b) and the backward through its frames: contraction as WRITTEN ------------------------------
// hipcc's default (-ffp-contract=fast-honor-pragmas) lets the backend fuse a multiply with an add from ANOTHER statement, and
// whether it does depends on what the inliner put next to what: s1 = |t1|^2 rsq(.) followed by s1 + s2 became one fma where the
// frames were used straight away (the redo of parked rows) and stayed two operations where they passed through a branch (the
// tile kernels, a dense round) -- one ulp in a denominator, 1e-5 (near-reflections) to 1e-1 (ties) in dM.  Everything a hard
// row goes through therefore fuses only what one expression spells out (fma calls, a * b + c): a row's bits do not depend on
// the kernel, the instantiation or the wave-mates it met (tests/test_gpu_parity.py::test_parked_and_dense_hard_rows_*).
#pragma clang fp contract(on)
// Orthogonalise columns p and q by a plane rotation (p, q) <- (c p + s q, c q - s p).
// With d = |p|^2 - |q|^2, g = 2 p.q, h = sqrt(d^2 + g^2):  (c, s) = (d + sgn(d) h, g) normalised,
// i.e. tan(theta) = g / (d + sgn(d) h), |theta| <= pi/4 (up to a common sign of both new columns,
// which a one-sided sweep does not care about).  Two transcendentals, no division.
template <class T> __device__ __forceinline__ void rotate(V3<T> &p, V3<T> &q) {
    typedef Tr<T> R;
    const T al = dot(p, p), be = dot(q, q), ga = dot(p, q);
    const T d = al - be;
    const T g = ga + ga;
    const T gg = g * g;
    const T h = R::sqrt(R::fma(d, d, gg)) + R::splat(Consts<typename R::scalar>::delta);
    const T ae = d + R::copysign(h, d);
    const T rw = R::rsq(R::fma(ae, ae, gg));
    const T c = ae * rw, s = g * rw;
    const V3<T> np = mk<T>(R::fma(c, p.x, s * q.x), R::fma(c, p.y, s * q.y), R::fma(c, p.z, s * q.z));
    const V3<T> nq = mk<T>(R::fma(c, q.x, -(s * p.x)), R::fma(c, q.y, -(s * p.y)), R::fma(c, q.z, -(s * p.z)));
    p = np;
    q = nq;
}

// True if the predicate holds on any lane of the wave (the result is wave-uniform: a scalar branch).
__device__ __forceinline__ bool wave_any(bool p) { return hw::any_lane(p); }

// A unit vector orthogonal to the unit vector u: e_k x u, k = index of the smallest |u_k| (z first).
template <class T> __device__ __forceinline__ V3<T> any_perp(V3<T> u) {
    typedef Tr<T> R;
    typedef typename R::scalar S;
    const T ax = R::abs(u.x), ay = R::abs(u.y), az = R::abs(u.z), zero = R::splat(S(0));
    const typename R::mask mz = R::le(az, ax) & R::le(az, ay), my = R::le(ay, ax);
    const V3<T> w = sel<T>(mz, mk<T>(-u.y, u.x, zero), sel<T>(my, mk<T>(u.z, zero, -u.x), mk<T>(zero, -u.z, u.y)));
    return scale<T>(w, R::rsq(dot(w, w)));
}

template <class T> struct SignedSvd {
    V3<T> u1, u2, u3;   // columns of U' (right-handed)
    V3<T> v1, v2, v3;   // columns of V  (right-handed)
    T s1, s2, s3;       // s3 carries the sign; in units of the PRESCALED matrix
    T inv_scale;        // M_prescaled = M * 2^k ;  inv_scale = 2^k  (multiply gradients by it)
};

template <class T> __device__ __forceinline__ V3<typename Tr<T>::scalar> lane3(V3<T> v, int i) {
    return mk<typename Tr<T>::scalar>(Tr<T>::get(v.x, i), Tr<T>::get(v.y, i), Tr<T>::get(v.z, i));
}
template <class T> __device__ __forceinline__ void set_lane3(V3<T> &v, int i, V3<typename Tr<T>::scalar> s) {
    Tr<T>::set(v.x, i, s.x); Tr<T>::set(v.y, i, s.y); Tr<T>::set(v.z, i, s.z);
}

// m: row-major 3x3 (m[3*i+j]).  WANT_S: also fill s1,s2,s3 (backward needs them).
// SWEEPS fixed sweeps; ADAPT: plus one more when the wave-wide residual test fails.
// MAX_EXTRA: how many adaptive sweeps may follow (1 for float32: a fourth sweep reaches round-off).
template <bool WANT_S, class T, int SWEEPS = kSweeps, bool ADAPT = true, int MAX_EXTRA = 1>
__device__ __forceinline__ SignedSvd<T> signed_svd(const T (&m_in)[9]) {
    typedef Tr<T> R;
    typedef typename R::scalar S;
    typedef Consts<S> K;
    SignedSvd<T> o;
    // 1. exact power-of-two prescale: largest |entry| lands in [0.5, 1)
    T mx = R::max(R::max(R::abs(m_in[0]), R::abs(m_in[1])), R::abs(m_in[2]));
    mx = R::max(mx, R::max(R::max(R::abs(m_in[3]), R::abs(m_in[4])), R::abs(m_in[5])));
    mx = R::max(mx, R::max(R::max(R::abs(m_in[6]), R::abs(m_in[7])), R::abs(m_in[8])));
    const typename R::ivec ex = R::neg_frexp_exp(mx);    // 0 for mx == 0; finite for inf/NaN too
    const T sc = R::ldexp(R::splat(S(1)), ex);
    T m[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = m_in[i] * sc;     // exact: sc is a power of two
    o.inv_scale = sc;

    // 2. one-sided Jacobi on the columns
    V3<T> a0 = mk<T>(m[0], m[3], m[6]), a1 = mk<T>(m[1], m[4], m[7]), a2 = mk<T>(m[2], m[5], m[8]);
    rotate(a0, a1);
    rotate(a0, a2);
    rotate(a1, a2);
    // 2a. A matrix whose columns were orthogonal to begin with -- reflections and near-reflections, permutations, diagonal
    // matrices: what the fast path hands over in bulk when it is given such a batch -- is done after ONE sweep: all three
    // pairs are tested, a row that passes keeps these columns whatever its wave-mates need (per row, as everywhere), and
    // the remaining sweeps are skipped when no row of the wave needs them (wave-uniform).
    if (SWEEPS > 1) {
        const T e0 = dot(a0, a0), e1 = dot(a1, a1), e2 = dot(a2, a2);
        const T h01 = dot(a0, a1), h02 = dot(a0, a2), h12 = dot(a1, a2);
        const T t2 = R::splat(K::tol2);
        const typename R::mask early = R::le(h01 * h01, e0 * e1 * t2) & R::le(h02 * h02, e0 * e2 * t2) & R::le(h12 * h12, e1 * e2 * t2);
        if (R::wave_any(R::mnot(early))) {
            const V3<T> f0 = a0, f1 = a1, f2 = a2;
#pragma unroll
            for (int sweep = 1; sweep < SWEEPS; ++sweep) {
                rotate(a0, a1);
                rotate(a0, a2);
                rotate(a1, a2);
            }
            a0 = sel<T>(early, f0, a0); a1 = sel<T>(early, f1, a1); a2 = sel<T>(early, f2, a2);
        }
    }

    // 2b. convergence test.  A cyclic sweep ends with rotation (1,2), which leaves gamma_12 = 0 and
    // gamma_02 = -s_12 * gamma_01: the (0,1) residual dominates, so it alone is tested:
    //     gamma_01^2  <=  tol^2 |a_0|^2 |a_1|^2 .
    T n0 = dot(a0, a0), n1 = dot(a1, a1), n2 = dot(a2, a2);
    if (ADAPT) {
        int extra = 0;
        while (true) {
            const T g01 = dot(a0, a1);
            const typename R::mask need = R::gt(g01 * g01, n0 * n1 * R::splat(K::tol2));   // NaN / zero rows compare false
            if (!R::wave_any(need)) break;
            // The branch is wave-uniform, the update is per matrix: a row keeps its three-sweep columns unless IT failed
            // the test, so its result does not depend on which other rows happen to share the wave.
            // float32: one more rotation of the pair that was tested is enough -- the other two residuals are already
            // second order (gamma_12 = 0, gamma_02 = -s_12 gamma_01), and on 16 adversarial families the repaired rows end
            // up as close to float64 LAPACK as after a full fourth sweep (|dR| gap/s1 <= 4.5e-7 either way).
            // float64 (MAX_EXTRA > 1) keeps whole sweeps: it iterates down to 1e-14.
            V3<T> b0 = a0, b1 = a1, b2 = a2;
            rotate(b0, b1);
            if (MAX_EXTRA > 1) {
                rotate(b0, b2);
                rotate(b1, b2);
            }
            a0 = sel<T>(need, b0, a0); a1 = sel<T>(need, b1, a1); a2 = sel<T>(need, b2, a2);
            n0 = dot(a0, a0); n1 = dot(a1, a1); n2 = dot(a2, a2);
            if (MAX_EXTRA == 1 || ++extra >= MAX_EXTRA) break;      // (compile-time exit for the float32 kernels)
        }
    }

    // 3. z = the smallest column (ties, i.e. singular values equal to within a few ulp as in diag(1,1,-1), go to the
    // LAST column as LAPACK's ordering does: the reference then maps a pure reflection to the identity); of the other
    // two the LARGER one is x.  The order matters on the V side: v1 = M^T u1 / s_x is exact to eps s1 / s_x, and v2 is
    // orthogonalised against it -- with a numerically zero x first (rank-one input) v1 would be noise and would drag
    // the one meaningful direction with it.  R does not depend on the order of the pair (u3 and v3 flip together);
    // s3' = u3 . z does, through the handedness of (x, y, z): `swapped` undoes it.
    // Three comparisons give a total preorder (column 2 enters them shrunk by the tie factor, so it loses near-ties):
    //   a column is the largest if it wins both of its comparisons and the middle one if it wins exactly one.
    const T n2t = n2 * R::splat(K::tie);
    const typename R::mask c01 = R::ge(n0, n1), c02 = R::ge(n0, n2t), c12 = R::ge(n1, n2t);
    const typename R::mask nc01 = R::mnot(c01);
    const typename R::mask x0 = c01 & c02, x1 = nc01 & c12;                      // else column 2
    const typename R::mask y0 = c01 ^ c02, y1 = R::mnot(c01 ^ c12);              // else column 2
    const V3<T> x = sel<T>(x0, a0, sel<T>(x1, a1, a2));
    const V3<T> y = sel<T>(y0, a0, sel<T>(y1, a1, a2));
    const T nx = R::sel(x0, n0, R::sel(x1, n1, n2));
    // (x, y, z) is an odd permutation of the columns for (1,0,2), (0,2,1), (2,1,0)
    const typename R::mask z0m = nc01 & R::mnot(c02), z2m = c02 & c12;
    const typename R::mask swapped = (x1 & y0) | (x0 & R::mnot(y0 | y1)) | (R::mnot(x0 | x1) & y1);
    const V3<T> z = sel<T>(z2m, a2, sel<T>(z0m, a0, a1));

    const T inx = R::rsq(nx);
    V3<T> u1 = scale<T>(x, inx);
    const T qy = dot(u1, y);
    V3<T> w = axpy<T>(-qy, u1, y);
    T nw = dot(w, w);
    const T inw = R::rsq(nw);
    V3<T> u2 = scale<T>(w, inw);
    const V3<T> mr0 = mk<T>(m[0], m[1], m[2]), mr1 = mk<T>(m[3], m[4], m[5]), mr2 = mk<T>(m[6], m[7], m[8]);
    V3<T> t1 = axpy<T>(u1.z, mr2, axpy<T>(u1.y, mr1, scale<T>(mr0, u1.x)));     // M^T u1 = s1 v1
    V3<T> t2 = axpy<T>(u2.z, mr2, axpy<T>(u2.y, mr1, scale<T>(mr0, u2.x)));     // M^T u2 = s2 v2
    // (|M^T u1| = |x| and |GS(M^T u2)| = |w| up to the square of the sweep residual, but reusing 1/|x| and 1/|w| here breaks
    // orthogonality for s2/s1 < 1e-2: the norms are recomputed.)
    T nt1 = dot(t1, t1);
    V3<T> v1 = scale<T>(t1, R::rsq(nt1));
    const T pt = dot(v1, t2);
    V3<T> r2 = axpy<T>(-pt, v1, t2);
    T nr2 = dot(r2, r2);
    V3<T> v2 = scale<T>(r2, R::rsq(nr2));

    // Rank <= 1 (or all-zero) input: the frame is not unique; pick one deterministically.  Numerically rank one
    // counts too: when the second singular value sits at round-off (s2 <~ eps s1; an outer product, a matrix of small
    // integers, nine equal network outputs), y or M^T u2 is noise that may lie ALONG the first vector, and one
    // Gram-Schmidt step then leaves a "unit vector" that is not orthogonal to it.  Such a step is recognised by what
    // it removed: |w|^2 <= 1e-2 (u1.y)^2, i.e. more than 90 % of the vector was parallel (what is left after a milder
    // step is orthogonal to ~10 eps).
    // (`<=` comparisons are false for NaN, so NaN input flows through the fast path to NaN output.)
    const T tiny = R::splat(K::tiny);
    const T lost = R::splat(S(1e-2));
    const typename R::mask degenerate = R::le(nx, tiny) | R::le(nw, R::fma(lost * qy, qy, tiny)) | R::le(nt1, tiny)
                                        | R::le(nr2, R::fma(lost * pt, pt, tiny));
    if (__builtin_expect(R::any(degenerate), 0)) {
        // (generic over T like everything else: round 2 looped over the halves of a packed pair with scalar code under a divergent
        // branch -- a batch of rank-one rows paid 10 % of its time there)
        const typename R::mask b0 = R::ge(n0, n1) & R::ge(n0, n2), b1 = R::ge(n1, n2);
        const V3<T> big = sel<T>(b0, a0, sel<T>(b1, a1, a2));
        const T nb = R::sel(b0, n0, R::sel(b1, n1, n2));
        const typename R::mask null = R::le(nb, tiny);                    // M == 0  ->  identity (matches the reference)
        const V3<T> ex = mk<T>(R::splat(S(1)), R::splat(S(0)), R::splat(S(0)));
        const V3<T> su1 = sel<T>(null, ex, scale<T>(big, R::rsq(nb)));
        const V3<T> st1 = axpy<T>(su1.z, mr2, axpy<T>(su1.y, mr1, scale<T>(mr0, su1.x)));
        const V3<T> sv1 = sel<T>(null, ex, scale<T>(st1, R::rsq(dot(st1, st1))));
        u1 = sel<T>(degenerate, su1, u1);
        v1 = sel<T>(degenerate, sv1, v1);
        u2 = sel<T>(degenerate, any_perp<T>(su1), u2);
        v2 = sel<T>(degenerate, any_perp<T>(sv1), v2);
        if (WANT_S) { nt1 = R::sel(degenerate, nb, nt1); nr2 = R::sel(degenerate, R::splat(S(0)), nr2); }
    }
    o.u1 = u1; o.u2 = u2; o.u3 = cross<T>(u1, u2);
    o.v1 = v1; o.v2 = v2; o.v3 = cross<T>(v1, v2);
    if (WANT_S) {
        // s_k = u_k^T M v_k; cheaper: |M^T u_k| for k = 1,2 and u3 . z for the signed one.
        o.s1 = nt1 * R::rsq(R::max(nt1, tiny));
        o.s2 = nr2 * R::rsq(R::max(nr2, tiny));
        const T s3u = dot(o.u3, z);
        o.s3 = R::sel(swapped, -s3u, s3u);
    } else {
        o.s1 = o.s2 = o.s3 = R::splat(S(0));
    }
    return o;
}

// R = U' V^T, row-major.  Rows 0 and 1 come from the frames (only the x,y components of u3 = u1 x u2 are
// needed for them); row 2 of a rotation is row0 x row1.
template <class T> __device__ __forceinline__ void rotation_rows(V3<T> u1, V3<T> u2, V3<T> v1, V3<T> v2, T (&r)[9]) {
    typedef Tr<T> R;
    const T u3x = R::fma(u1.y, u2.z, -(u1.z * u2.y));
    const T u3y = R::fma(u1.z, u2.x, -(u1.x * u2.z));
    const V3<T> v3 = cross<T>(v1, v2);
    const V3<T> r0 = axpy<T>(u3x, v3, axpy<T>(u2.x, v2, scale<T>(v1, u1.x)));
    const V3<T> r1 = axpy<T>(u3y, v3, axpy<T>(u2.y, v2, scale<T>(v1, u1.y)));
    const V3<T> r2 = cross<T>(r0, r1);
    r[0] = r0.x; r[1] = r0.y; r[2] = r0.z;
    r[3] = r1.x; r[4] = r1.y; r[5] = r1.z;
    r[6] = r2.x; r[7] = r2.y; r[8] = r2.z;
}
template <class T> __device__ __forceinline__ void rotation_from(const SignedSvd<T> &f, T (&r)[9]) {
    rotation_rows<T>(f.u1, f.u2, f.v1, f.v2, r);
}

#pragma clang fp contract(fast)

// =====================================================================================================================
// K1 forward, fast path: the rotation as the dominant eigenvector of Davenport's 4x4 matrix.
//
// R = argmax_{R in SO(3)} tr(R^T M) is exactly U diag(1,1,det(UV^T)) V^T (rotation_representation.py:199-205), and with
// R = R(q), q a unit quaternion (w,x,y,z), tr(R^T M) = q^T K q for the symmetric traceless
//     K = [ tr M          m21-m12        m02-m20        m10-m01      ]
//         [  .         m00-m11-m22       m01+m10        m02+m20      ]
//         [  .             .          -m00+m11-m22      m12+m21      ]
//         [  .             .              .          -m00-m11+m22    ]
// whose eigenvalues are s1+s2+s3', s1-s2-s3', -s1+s2-s3', -s1-s2+s3' (s3' = det-signed): the gap between the two largest,
// 2(s2+s3'), is the conditioning of R itself.  Per matrix:
//   1. (the method is homogeneous in M: only rows far from unit scale are prescaled, by a power of two, in a rare branch);
//   2. lambda_max = largest root of  l^4 - 2|M|^2 l^2 - 8 det(M) l + (|M|^4 - 4|cof M|^2):  started at s1 + s2 + s3' with the
//      singular values from the closed-form roots of the cubic of M^T M (good to 5e-6 on the median row), then two Newton steps;
//   3. q = the largest column of adj(lambda I - K) = (product of the three gaps) q q^T -- no division, no pivoting;
//   4. lam2 = Rayleigh quotient of q (error squared).  The quotient's distance from the shift MEASURES the shift's error: a row whose
//      quotient stays within kQuatClose of it keeps q; the others take q again from the adjugate at the quotient, in one rare loop;
//   5. R(q) -- orthogonal by construction.
// Per PAIR of matrices (round 5, K1's counters): 340 vector instructions -- about 215 packed, 26 transcendental, 100 plain -- against
// 569 (394 / 45 / 130) for the three Jacobi sweeps and the frames above (rounds 2-4: 408-431; docs/history/tools.tar.gz:tools/proto/qpath2.py is the numpy float32
// prototype: 1M Gaussian rows median |dR| 1.4e-7, |dR| gap/s1 <= 1.4e-6 on twenty adversarial families).
//
// What the fast path cannot do it says so: a row is HARD when (a) the Rayleigh quotient moved lambda by more than kQuatConv times a lower
// bound of the gap (the root finder had not converged, or the gap product tr adj(lambda I - K) is too small for the adjugate's column to
// be signal: with the 2-ulp floor on the move the same test says tr adj >= 1.2e-3 lambda^3 -- ill-conditioned, rank-deficient, ties:
// everything where the reference's answer is a matter of LAPACK's ordering), or (b) lambda is not certified as the LARGEST root
// (quat_settled), or (c) it is hard by its invariants or anything is not finite.  Hard rows (2e-6 of Gaussian input) are redone by the
// Jacobi path above, one row at a time, by the caller -- a row's result never depends on its wave-mates.
constexpr float kQuatTau2 = 1e-5f;      // a gap product below half of this (times lambda^3) cannot be helped by refining lambda (see `hopeless`)
constexpr int kQuatExtra = 3;           // how many refinements of (lambda, q) a row may take
#ifndef SO3_QUAT_CONV
#define SO3_QUAT_CONV 4e-4f
#endif
constexpr float kQuatConv = SO3_QUAT_CONV;
constexpr float kQuatUlps = 1.2e-7f;    // 2 ulp: the round-off of a float32 Rayleigh quotient, added to every measured move of lambda
#ifndef SO3_QUAT_CLOSE
#define SO3_QUAT_CLOSE 0.9e-6f
#endif
constexpr float kQuatClose = SO3_QUAT_CLOSE;     // a first eigenvector whose Rayleigh quotient lies within this times s1 of its shift is final
constexpr float kQuatWindowLo = 3.7252903e-9f, kQuatWindowHi = 17179869184.0f;   // 2^-28 <= |M|_F^2 <= 2^34 (see quat_rotation, step 1)
// (0.5 through round 5's last day: the adversarial search then found accepted rows at 2.0-2.2e-6 under seeds other than the one the test
// used -- neither kQuatConv nor kQuatClose moves that number, the second gap does.  Worst over 20-60 seeds x 2e7 rows: 0.8 -> 1.90e-6,
// 1.0 -> 1.94e-6 (fifty seeds; 1.82e-6 over another forty), 1.3 -> 1.70e-6, 2.0 -> 1.63e-6; Gaussian rows that turn hard: 2e-6, 3.5e-6,
// 1.4e-5, more -- and hard rows are not free even when they are a handful per million (their workgroups redo them behind the loop: one
// Jacobi's latency): K1 as a graph 14.00 (0.5), 14.01 (0.8), 14.15 (1.0), 14.65 us (1.3).  tools/search_seeds.py)
#ifndef SO3_QUAT_CURV
#define SO3_QUAT_CURV 0.8f
#endif
constexpr float kQuatCurv = SO3_QUAT_CURV;      // P''(lambda) >= kQuatCurv |M|^2: the SECOND gap is not small either (see quat_settled)

#ifdef SO3_HOST_MODEL
// What the CPU suite counts while it drives these templates (oracle/kernel_model.cpp): deterministic stand-ins for "how often does
// a wave take the rare branch", which on the device is a matter of timing runs.
struct HostCounters {
    long long refined_rows = 0;      // rows whose first eigenvector was not final (quat_rotation_core, step 6)
    long long refinements = 0;       // adjugates computed for them
};
inline HostCounters &host_counters() { static HostCounters c; return c; }
#endif

template <class T> struct Sym4 {        // symmetric 4x4, upper triangle
    T a00, a01, a02, a03, a11, a12, a13, a22, a23, a33;
};

// The largest column of adj(lam I - K) and the trace of the adjugate.
template <class T> struct Quat {        // (w, x, y, z), unnormalised.  Four named members, not T[4]: the compiler made ONE 256-bit value of the array,
    T w, x, y, z;                       // and where the rare refinement branch rejoins the round it then copied all of it, whichever path had been taken
};
template <class T> __device__ __forceinline__ void dominant_column(const Sym4<T> &k, T lam, Quat<T> &q, T &trace) {
    typedef Tr<T> R;
    const T n00 = lam - k.a00, n11 = lam - k.a11, n22 = lam - k.a22, n33 = lam - k.a33;
    const T n01 = -k.a01, n02 = -k.a02, n03 = -k.a03, n12 = -k.a12, n13 = -k.a13, n23 = -k.a23;
    // 2x2 minors of rows (0,1) and of rows (2,3); s5 = c0 by symmetry
    const T s0 = R::fma(n00, n11, -(n01 * n01)), s1 = R::fma(n00, n12, -(n01 * n02)), s2 = R::fma(n00, n13, -(n01 * n03));
    const T s3 = R::fma(n01, n12, -(n11 * n02)), s4 = R::fma(n01, n13, -(n11 * n03)), s5 = R::fma(n02, n13, -(n12 * n03));
    const T c5 = R::fma(n22, n33, -(n23 * n23)), c4 = R::fma(n12, n33, -(n13 * n23)), c3 = R::fma(n12, n23, -(n13 * n22));
    const T c2 = R::fma(n02, n33, -(n03 * n23)), c1 = R::fma(n02, n23, -(n03 * n22));
    const T b00 = R::fma(n13, c3, R::fma(n11, c5, -(n12 * c4)));
    const T b01 = R::fma(n02, c4, -R::fma(n03, c3, n01 * c5));
    const T b02 = R::fma(n33, s3, R::fma(n13, s5, -(n23 * s4)));
    const T b03 = R::fma(n22, s4, -R::fma(n23, s3, n12 * s5));
    const T b11 = R::fma(n03, c1, R::fma(n00, c5, -(n02 * c2)));
    const T b12 = R::fma(n23, s2, -R::fma(n33, s1, n03 * s5));
    const T b13 = R::fma(n23, s1, R::fma(n02, s5, -(n22 * s2)));
    const T b22 = R::fma(n33, s0, R::fma(n03, s4, -(n13 * s2)));
    const T b23 = R::fma(n12, s2, -R::fma(n23, s0, n02 * s4));
    const T b33 = R::fma(n22, s0, R::fma(n02, s3, -(n12 * s1)));
    trace = (b00 + b11) + (b22 + b33);
    // adj = (g2 g3 g4) q q^T: its diagonal is proportional to q_j^2 -- take the column of the largest
    const typename R::mask m01 = R::ge(b00, b11), m23 = R::ge(b22, b33);
    const T da = R::sel(m01, b00, b11), db = R::sel(m23, b22, b33);
    const T a0 = R::sel(m01, b00, b01), a1 = R::sel(m01, b01, b11), a2 = R::sel(m01, b02, b12), a3 = R::sel(m01, b03, b13);
    const T e0 = R::sel(m23, b02, b03), e1 = R::sel(m23, b12, b13), e2 = R::sel(m23, b22, b23), e3 = R::sel(m23, b23, b33);
    const typename R::mask mab = R::ge(da, db);
    q.w = R::sel(mab, a0, e0); q.x = R::sel(mab, a1, e1); q.y = R::sel(mab, a2, e2); q.z = R::sel(mab, a3, e3);
}

// Rayleigh quotient lambda = q^T K q / q^T q; inv_n = 1 / q^T q (R(q) needs it again).
template <class T> __device__ __forceinline__ T norm2(const Quat<T> &q) {
    typedef Tr<T> R;
    return R::fma(q.z, q.z, R::fma(q.y, q.y, R::fma(q.x, q.x, q.w * q.w)));
}
template <class T> __device__ __forceinline__ T rayleigh(const Sym4<T> &k, const Quat<T> &q, T &inv_n) {
    typedef Tr<T> R;
    const T kq0 = R::fma(k.a03, q.z, R::fma(k.a02, q.y, R::fma(k.a01, q.x, k.a00 * q.w)));
    const T kq1 = R::fma(k.a13, q.z, R::fma(k.a12, q.y, R::fma(k.a11, q.x, k.a01 * q.w)));
    const T kq2 = R::fma(k.a23, q.z, R::fma(k.a22, q.y, R::fma(k.a12, q.x, k.a02 * q.w)));
    const T kq3 = R::fma(k.a33, q.z, R::fma(k.a23, q.y, R::fma(k.a13, q.x, k.a03 * q.w)));
    const T num = R::fma(q.z, kq3, R::fma(q.y, kq2, R::fma(q.x, kq1, q.w * kq0)));
    inv_n = R::rcp(norm2<T>(q));
    return num * inv_n;
}

// A row is settled when (1) the Rayleigh quotient moved lambda by at most kQuatConv times the gap's lower bound
// trace / (2 lambda)^2 -- trace = tr adj(lambda I - K) = P'(lambda), the product of the three gaps -- and (2) lambda is the LARGEST
// root: with P' > 0 (from 1), P'' > 0 and the third derivative 24 lambda > 0 the Budan-Fourier count allows at most one root above
// lambda.  (Without 2 an exact double root at the top -- small-integer matrices with s2 = s3 and det < 0 -- could throw the
// iteration below it and on to the third eigenvalue, whose adjugate looks just as healthy.)  (2) is asked with a margin,
// P'' >= kQuatCurv |M|_F^2, criterion (3): with gaps g2 <= g3 <= g4 of lambda to the other eigenvalues,
// P''/2 = g2 g3 + g2 g4 + g3 g4 <= 3 g3 g4, so the margin bounds the SECOND gap g3 = 2 (s1 + s3') from below.  The adjugate's
// round-off is eps g4^3 / (g2 g3 g4): a small g2 is the conditioning of R itself, a small g3 (all three singular values close
// and det < 0: a near-reflection) is a weakness of the quaternion formulation only, and such rows go to the Jacobi path.
// The scale of (1) is L = lam_after + |move| >= both values (a settled row's lambda is positive): a shift far above the spectrum
// has a huge, healthy-looking adjugate.  The measured move carries a floor of 2 ulp of L: lambda itself has that much round-off
// however still the iteration stands (a move of exactly zero proved nothing: round 3's search on the device found rows with a gap
// of 7e-6 s1 accepted that way, their rotation off by 0.7).  With the floor, (1) also says trace >= 4 kQuatUlps / kQuatConv L^3 =
// 1.2e-3 L^3 -- which is why rounds 2-4's separate bar "trace > 1e-3 lambda^3" is gone: it was implied.
// `move` (out): |lam_before - lam_after|, for the caller's own use.
template <class T>
__device__ __forceinline__ typename Tr<T>::mask quat_settled(T lam_before, T lam_after, T trace, T twoc2, T f, T &move) {
    typedef Tr<T> R;
    typedef typename R::scalar S;
    move = R::abs(lam_before - lam_after);
    const T big = lam_after + move;
    const T moved = R::fma(big, R::splat(S(kQuatUlps)), move);
    const typename R::mask converged = R::le(moved * ((big * big) * R::splat(S(4))), trace * R::splat(S(kQuatConv)));
    const typename R::mask topmost = R::gt(R::fma(R::splat(S(12)) * lam_after, lam_after, twoc2), f * R::splat(S(kQuatCurv))) & R::gt(lam_after, R::splat(S(0)));
    return converged & topmost;
}

// The exact power of two a row outside the fast path's scale window was multiplied by (1 elsewhere): the backward from the
// rotation works on the same prescaled matrix.
template <class T> struct Prescale {
    T factor;
    bool any;          // wave-uniform: some row of the wave has factor != 1
};
// The fast path on a matrix whose |M|_F^2 = f is known; in_window: the rows whose f lies inside the scale window (the others are
// declared hard at the end -- after the prescale these are zero, infinite and NaN rows).
// The head of the fast path: cofactors, det, |adj M|_F^2 and the cubic of the squared singular values in units of f = |M|_F^2
// (mu / f in [0, 1]: no power of the entries beyond f^2 is formed).  Shared by quat_rotation_core and by invariant_hard_rows, so
// that both judge a row by the same bits.
template <class T> struct CubicHead {
    T det, cf;      // det M, |adj M|_F^2 = s1^2 s2^2 + s1^2 s3^2 + s2^2 s3^2
    T cfn, dn;      // cf / f^2,  det^2 / f^3
    T p, x;         // p = (1 - 3 cfn)/9 (floored at 1e-12),  x = cos(theta) = q / p^(3/2) clamped to [-1, 1]
};
template <class T> __device__ __forceinline__ CubicHead<T> cubic_head(const T (&m)[9], T f) {
    typedef Tr<T> R;
    typedef typename R::scalar S;
    CubicHead<T> h;
    const T g00 = R::fma(m[4], m[8], -(m[5] * m[7])), g01 = R::fma(m[5], m[6], -(m[3] * m[8])), g02 = R::fma(m[3], m[7], -(m[4] * m[6]));
    const T g10 = R::fma(m[2], m[7], -(m[1] * m[8])), g11 = R::fma(m[0], m[8], -(m[2] * m[6])), g12 = R::fma(m[1], m[6], -(m[0] * m[7]));
    const T g20 = R::fma(m[1], m[5], -(m[2] * m[4])), g21 = R::fma(m[2], m[3], -(m[0] * m[5])), g22 = R::fma(m[0], m[4], -(m[1] * m[3]));
    h.det = R::fma(m[2], g02, R::fma(m[1], g01, m[0] * g00));
    T cf = g00 * g00;
    cf = R::fma(g01, g01, cf); cf = R::fma(g02, g02, cf); cf = R::fma(g10, g10, cf); cf = R::fma(g11, g11, cf);
    cf = R::fma(g12, g12, cf); cf = R::fma(g20, g20, cf); cf = R::fma(g21, g21, cf); cf = R::fma(g22, g22, cf);
    h.cf = cf;
    const T inv_f = R::rcp(f);
    const T di = h.det * inv_f;
    h.cfn = (cf * inv_f) * inv_f;
    h.dn = (di * di) * inv_f;
    h.p = R::max(R::fma(h.cfn, R::splat(S(-1.0 / 3.0)), R::splat(S(1.0 / 9.0))), R::splat(S(1e-12)));
    const T q = R::fma(h.dn, R::splat(S(0.5)), R::fma(h.cfn, R::splat(S(-1.0 / 6.0)), R::splat(S(1.0 / 27.0))));
    h.x = R::clamp(q * R::rsq((h.p * h.p) * h.p), R::splat(S(-1)), R::splat(S(1)));
    return h;
}
// Rows that are hard by their invariants alone: no eigenvalue is needed to see that the three singular values coincide with
// det < 0 (a near-reflection: 27 det^2 = |M|^6 is the equality case of the AM-GM inequality) or that the rank is one
// (|adj M| = 0).  Each bar lies well inside what steps 4b-6 call hard anyway (none in 2e7 Gaussian rows); the point is that EVERY
// row of such a family is caught -- Newton's lambda, on which 4b judges, is still far from a multiple root on 2-7 % of them -- so
// that a round made of them can be recognised before the fast path is run (OpProject's dense-round shortcut).  A per-row rule
// like every other: it is folded into the verdict below.  (Ties s2 = s3 with det < 0 show as x = +1, the cubic's two smaller
// roots coinciding, but float32 resolves x to 1e-7, i.e. the tie to 5e-4: a bar loose enough to catch every row of a batch of
// ties -- x >= 1 - 1e-5 did not -- already takes 1e-4 of the Gaussian rows, and one queued row per 30 waves lengthens the
// launch's tail by a Jacobi pass: K1 8 % slower on Gaussian input.  No rule for ties.)
template <class T> __device__ __forceinline__ typename Tr<T>::mask invariant_hard(const CubicHead<T> &h) {
    typedef Tr<T> R;
    typedef typename R::scalar S;
    const typename R::mask refl = R::gt(R::splat(S(0)), h.det) & R::ge(h.dn * R::splat(S(27)), R::splat(S(1.0 - 1e-4)));
    const typename R::mask rank1 = R::le(h.cfn, R::splat(S(1e-12)));
    return refl | rank1;
}

template <class T, bool SKIP = true> __device__ __forceinline__ typename Tr<T>::mask quat_rotation_core(const T (&m)[9], T f, typename Tr<T>::mask in_window, T (&r)[9]) {
    typedef Tr<T> R;
    typedef typename R::scalar S;
    // 2. K (order w, x, y, z)
    const T tr = (m[0] + m[4]) + m[8];
    const T two = R::splat(S(2));
    Sym4<T> k;
    k.a00 = tr; k.a11 = R::fma(two, m[0], -tr); k.a22 = R::fma(two, m[4], -tr); k.a33 = R::fma(two, m[8], -tr);
    k.a01 = m[7] - m[5]; k.a02 = m[2] - m[6]; k.a03 = m[3] - m[1];
    k.a12 = m[1] + m[3]; k.a13 = m[2] + m[6]; k.a23 = m[5] + m[7];
    // 3. the characteristic quartic  l^4 + c2 l^2 + c1 l + c0
    const CubicHead<T> head = cubic_head<T>(m, f);
    const T det = head.det, cf = head.cf;
    // Rows that are hard whatever the steps below would say -- outside the scale window (zero, Inf, NaN) or hard by their invariants --
    // are judged here and folded into one mask: a per-row rule, so a row's result does not depend on its wave-mates.  A wave in which
    // EVERY row is such a row (SKIP) leaves for the Jacobi path at once: a batch of zeros (a dead head) costs K3 1.05 x a Gaussian
    // one instead of 1.34 x.  (K1's engine kernel asks the same question in front of the fast path, OpProject; inside it, the second
    // way out costs 20 registers that kernel does not have at three waves per SIMD.)
    const typename R::mask usable = in_window & R::mnot(invariant_hard<T>(head));
    if (SKIP && __builtin_expect(!R::wave_any(usable), 0)) return R::mnot(usable);      // r is not used for hard rows
    const T c2 = f * R::splat(S(-2)), c1 = det * R::splat(S(-8)), c0 = R::fma(f, f, cf * R::splat(S(-4)));
    const T twoc2 = c2 + c2;
    // 4. lambda_max = s1 + s2 + s3'.  Start: the squared singular values are the roots of  mu^3 - f mu^2 + cf mu - det^2,
    // in closed form  mu_k = f/3 + 2 sqrt(p) cos(theta/3 + 2 pi k/3),  p = (f^2 - 3 cf)/9,  cos(theta) = q / p^(3/2),
    // q = (2 f^3 - 9 f cf + 27 det^2)/54 -- with a four-term acos (7e-5), the hardware cosine and mu2 from the trace this
    // lands within 5e-6 of lambda_max on the median Gaussian row (p99 3e-4, worst 8e-3: small singular values come out
    // of a cancellation, but they weigh little in the sum).  Two Newton steps on the quartic finish it; the start is
    // raised by 1e-3 so that they come from above.  (Laguerre from the bound sqrt(3)|M|_F needed four steps and a Newton
    // step for the same roots: 68 packed instructions and 20 transcendentals against 40 and 26.)
    T lam, s1_start;       // s1_start: the largest singular value as the closed form gives it (1e-5 or better: a scale, see step 5)
    {
        const T p = head.p, x = head.x;
        const T ax = R::abs(x);
        T poly = R::fma(ax, R::splat(S(-0.0187293)), R::splat(S(0.0742610)));
        poly = R::fma(poly, ax, R::splat(S(-0.2121144)));
        poly = R::fma(poly, ax, R::splat(S(1.5707288)));
        const T acos_abs = R::sqrt(R::splat(S(1)) - ax) * poly;                                 // acos(|x|), radians
        // theta / 3 in revolutions: acos(x) = pi - acos(|x|) for x < 0
        const T third_rev = R::splat(S(1.0 / (6.0 * 3.14159265358979323846)));
        const T th = R::sel(R::ge(x, R::splat(S(0))), acos_abs * third_rev, R::fma(acos_abs, -third_rev, R::splat(S(1.0 / 6.0))));
        const T two_sp = R::sqrt(p) * R::splat(S(2)), third = R::splat(S(1.0 / 3.0));
        const T mu1 = R::fma(two_sp, R::cos_rev(th), third);
        const T mu3 = R::fma(two_sp, R::cos_rev(th + third), third);
        const T mu2 = (R::splat(S(1)) - mu1) - mu3;
        const T s3 = R::copysign(R::sqrt(R::abs(mu3)), det);
        const T r1 = R::sqrt(R::abs(mu1)), rf = R::sqrt(f);
        s1_start = r1 * rf;
        lam = ((r1 + R::sqrt(R::abs(mu2))) + s3) * (rf * R::splat(S(1.001)));
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const T l2 = lam * lam;
        const T p = R::fma(R::fma(l2 + c2, lam, c1), lam, c0);
        const T dp = R::fma(R::fma(R::splat(S(4)), l2, twoc2), lam, c1);
        lam = R::fma(-p, R::rcp(dp), lam);
    }
    // 5. eigenvector and its Rayleigh quotient lam2.  On Gaussian input the root is good to an ulp or two for all but 2e-3 of the
    // rows, and then the first vector is as good as a second one would be: with delta = lam - lambda the vector carries
    // eps = delta / g2 of its neighbour, the error of R in the measure it is judged by (|dR| gap / s1) is delta / s1, and the quotient,
    // being second-order accurate, MEASURES delta: a row whose quotient lies within kQuatClose s1 of its shift, and which is settled
    // (quat_settled above), keeps its first vector.  The quotient's own round-off, 2 ulp of LAMBDA, is charged to the move, so that the
    // bar means the same whatever lambda / s1 is (1 to 3).  What the bar buys, measured: 1.6e-3 of Gaussian rows refine (one round of 128
    // in five or six) and the device search (tests/test_gpu_certificate_search.py, 2e7 bred rows) finds a worst accepted |dR| gap / s1
    // of 1.6e-6 against its bound (2e-6 then, 2.5e-6 since round 6: sixty seeds read 1.90e-6) (round 4's residual test: 1.7e-3 and 1.25e-6; a bar of 1.1e-6: 1.0e-3 and 1.7e-6).  Where the
    // error of the first vector comes from is not Newton -- a third step changed nothing -- but the quartic's float32 coefficients,
    // which move its root by eps lambda^4 / tr adj: 1e-6 lambda where the gap product is a third of lambda^3.
    // (Rounds 2-4 asked the residual |K q - lam2 q| instead: 13 packed instructions per pair for the same decision -- what else the
    // residual sees, the adjugate's own round-off, a second vector has too.)
    Quat<T> q;
    T trace, inv_n, move;
    dominant_column<T>(k, lam, q, trace);
    const T lam2 = rayleigh<T>(k, q, inv_n);
    typename R::mask settled = quat_settled<T>(lam, lam2, trace, twoc2, f, move);
    settled = settled & R::le(R::fma(lam2, R::splat(S(kQuatUlps)), move), s1_start * R::splat(S(kQuatClose)));
    // 6. rows this did not settle (every comparison is written so that NaN makes the row unsettled) are refined -- q again from the
    // adjugate at the quotient, which squares the error -- under ONE wave-uniform branch (one round of 128 rows in five on Gaussian
    // input), in a loop that its own rows keep running (a second pass one round in sixty, a third one in five hundred): a refined
    // vector is judged by how far the quotient of its predecessor lay from the shift the predecessor was computed at.  That settles
    // rows whose root was still on its way and rows with a gap down to ~3e-4 of lambda; what is left (rank-deficient, ties, gaps at
    // round-off) is hard.  A row whose gap product does not reach half of kQuatTau2 lambda^3, or whose curvature P'' does not
    // reach half of criterion (3)'s bar (a near-reflection), cannot be settled by refining lambda (neither moves by a factor of
    // two): it is FROZEN -- it takes no refinement and does not hold its wave in the loop -- and goes to the Jacobi path, so that
    // batches of ties, reflections and rank-deficient rows pay the fast path once.  Per row, like everything here: a settled or
    // frozen row never takes a refinement that a wave-mate asked for.
    // (rows that are hard whatever happens -- outside the window, hard by their invariants -- do not ask for it: with 1 % of reflections or
    // rank-one rows in a batch every round of 128 holds one, and paid an adjugate for a row whose verdict was in already)
    if (__builtin_expect(R::wave_any(R::mnot(settled) & usable), 0)) {
#ifdef SO3_HOST_MODEL
        ++host_counters().refined_rows;          // (one "lane" per row on the host: how often the device's wave-uniform branch would be asked for)
#endif
        // (judged again at every refined lambda: at a double root Newton's lambda is still 1e-3 away and the gap product looks fine there;
        // the quotient then halves the distance per pass and never settles -- a batch of ties or rank-one rows paid all three passes)
        auto cannot_settle = [&](T tr_, T l_) {
            const T ll = l_ * l_;
            return R::mnot(R::gt(tr_, (ll * l_) * R::splat(S(0.5f * kQuatTau2))) & R::gt(R::fma(R::splat(S(12)), ll, twoc2), f * R::splat(S(0.5f * kQuatCurv))));
        };
        typename R::mask hopeless = cannot_settle(trace, lam2) | R::mnot(usable);       // (a row that is hard by its invariants does not hold its wave either)
        typename R::mask frozen = settled | hopeless;
        T shift = lam, quot = lam2;              // the shift the current q was computed at, and q's Rayleigh quotient
#pragma unroll 1
        for (int extra = 0;; ++extra) {
            Quat<T> qn;
            T tracen, unused;
#ifdef SO3_HOST_MODEL
            ++host_counters().refinements;
#endif
            dominant_column<T>(k, quot, qn, tracen);
            const typename R::mask good = quat_settled<T>(shift, quot, tracen, twoc2, f, unused);
            q.w = R::sel(frozen, q.w, qn.w); q.x = R::sel(frozen, q.x, qn.x); q.y = R::sel(frozen, q.y, qn.y); q.z = R::sel(frozen, q.z, qn.z);
            shift = R::sel(frozen, shift, quot);
            settled = settled | (good & R::mnot(frozen));
            hopeless = hopeless | cannot_settle(tracen, quot);
            frozen = settled | hopeless;
            if (extra + 1 >= kQuatExtra || !R::wave_any(R::mnot(frozen))) break;
            quot = rayleigh<T>(k, q, inv_n);
        }
        inv_n = R::rcp(norm2<T>(q));
    }
    // 7. R(q), q = (w, x, y, z) unnormalised
    const T s2 = inv_n + inv_n;
    const T w = q.w, x = q.x, y = q.y, z = q.z;
    const T xs = x * s2, ys = y * s2, zs = z * s2;
    const T wx = w * xs, wy = w * ys, wz = w * zs;
    const T one = R::splat(S(1));
    const T dz = R::fma(-z, zs, one), dy = R::fma(-y, ys, one);
    r[0] = R::fma(-y, ys, dz); r[1] = R::fma(x, ys, -wz); r[2] = R::fma(x, zs, wy);
    r[3] = R::fma(x, ys, wz); r[4] = R::fma(-x, xs, dz); r[5] = R::fma(y, zs, -wx);
    r[6] = R::fma(x, zs, -wy); r[7] = R::fma(y, zs, wx); r[8] = R::fma(-x, xs, dy);
    const typename R::mask finite = R::le(R::abs(s2), R::splat(S(3e38)));
    return R::mnot(settled & finite & usable);
}

// r = the rotation nearest to m_in (fast path); returns the mask of HARD rows, whose r must not be used.
// `prescale` (optional): the exact power of two a row outside the scale window was multiplied by (1 elsewhere).
template <class T, bool SKIP = true> __device__ __forceinline__ typename Tr<T>::mask quat_rotation(const T (&m_in)[9], T (&r)[9], Prescale<T> *prescale = nullptr) {
    typedef Tr<T> R;
    typedef typename R::scalar S;
    // 1. scale.  Every step of the core is homogeneous in M (K and lambda scale with M, the adjugate with its cube, all the tests
    // compare like with like), so the fast path works on the matrix as it comes as long as |M|_F^2 stays within
    // [2^-28, 2^34] (the Rayleigh quotient's numerator lambda |q|^2 ~ 64 lambda^7 must stay finite -- rows between 2^34.4 and
    // 2^36, inside round 2's window, came out hard -- and sixth powers of the entries a few orders clear of the underflow
    // threshold: entries between 2e-5 and 4e4).  Network outputs are O(1): no unconditional prescale (it cost
    // 9 packed and 16 plain instructions per pair of matrices), and the common path works on m_in itself, without a copy.
    // (row by row: three chains of three -- one chain of nine dependent packed instructions issues with a wait state between each two)
    const T f0 = R::fma(m_in[2], m_in[2], R::fma(m_in[1], m_in[1], m_in[0] * m_in[0]));
    const T f1 = R::fma(m_in[5], m_in[5], R::fma(m_in[4], m_in[4], m_in[3] * m_in[3]));
    const T f2 = R::fma(m_in[8], m_in[8], R::fma(m_in[7], m_in[7], m_in[6] * m_in[6]));
    const T f = (f0 + f1) + f2;
    const typename R::mask inside = R::ge(f, R::splat(S(kQuatWindowLo))) & R::le(f, R::splat(S(kQuatWindowHi)));
    const bool any_outside = R::wave_any(R::mnot(inside));
    if (prescale != nullptr) { prescale->factor = R::splat(S(1)); prescale->any = any_outside; }
    if (__builtin_expect(!any_outside, 1)) return quat_rotation_core<T, SKIP>(m_in, f, inside, r);
    // Rows outside the window get an exact power-of-two prescale (largest |entry| -> [0.5, 1); R does not depend on the scale)
    // under this wave-uniform branch; rows inside it are multiplied by 1 -- their bits do not change.  Zero, infinite and NaN
    // rows stay outside the window and are declared hard by the core.
    T m[9];
    T mx = R::max(R::max(R::abs(m_in[0]), R::abs(m_in[1])), R::abs(m_in[2]));
    mx = R::max(mx, R::max(R::max(R::abs(m_in[3]), R::abs(m_in[4])), R::abs(m_in[5])));
    mx = R::max(mx, R::max(R::max(R::abs(m_in[6]), R::abs(m_in[7])), R::abs(m_in[8])));
    const T sc = R::sel(inside, R::splat(S(1)), R::ldexp(R::splat(S(1)), R::neg_frexp_exp(mx)));
    if (prescale != nullptr) prescale->factor = sc;
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = m_in[i] * sc;
    const T fs0 = R::fma(m[2], m[2], R::fma(m[1], m[1], m[0] * m[0]));
    const T fs1 = R::fma(m[5], m[5], R::fma(m[4], m[4], m[3] * m[3]));
    const T fs2 = R::fma(m[8], m[8], R::fma(m[7], m[7], m[6] * m[6]));
    const T fs = (fs0 + fs1) + fs2;
    const typename R::mask inside2 = R::ge(fs, R::splat(S(kQuatWindowLo))) & R::le(fs, R::splat(S(kQuatWindowHi)));
    typename R::mask hard = quat_rotation_core<T, SKIP>(m, fs, inside2, r);
    // An all-zero row (a dead head) is the identity (the reference: the SVD of the zero matrix comes back with U = V = I), and the Jacobi
    // path gives exactly that -- so the forward says so here, in the branch such a row has taken anyway, instead of sending it
    // there: 1 % of zero rows cost K1 1.2-1.36 x a Gaussian batch.  Forward only (no `prescale`): the backward of a zero row goes
    // through the frames' floored denominators, which the rotation alone does not carry.
    if (prescale == nullptr) {
        const typename R::mask zero = R::le(mx, R::splat(S(0)));          // largest |entry| is 0; NaN compares false
        const T one = R::splat(S(1)), nil = R::splat(S(0));
#pragma unroll
        for (int i = 0; i < 9; ++i) r[i] = R::sel(zero, (i & 3) == 0 ? one : nil, r[i]);
        hard = hard & R::mnot(zero);
    }
    return hard;
}

// Does the wave hold nothing but rows that are hard by their invariants (and inside the scale window, where the core would
// judge them on the same bits)?  Wave-uniform; 60 instructions instead of the fast path's 420.
template <class T> __device__ __forceinline__ bool all_rows_invariant_hard(const T (&m)[9]) {
    typedef Tr<T> R;
    typedef typename R::scalar S;
    const T f0 = R::fma(m[2], m[2], R::fma(m[1], m[1], m[0] * m[0]));         // (the same bits as quat_rotation's)
    const T f1 = R::fma(m[5], m[5], R::fma(m[4], m[4], m[3] * m[3]));
    const T f2 = R::fma(m[8], m[8], R::fma(m[7], m[7], m[6] * m[6]));
    const T f = (f0 + f1) + f2;
    const typename R::mask inside = R::ge(f, R::splat(S(kQuatWindowLo))) & R::le(f, R::splat(S(kQuatWindowHi)));
    const typename R::mask sure = inside & invariant_hard<T>(cubic_head<T>(m, f));
    return !R::wave_any(R::mnot(sure));
}

// K1's arithmetic for every forward entry point: the fast path, and the Jacobi path for the rows it declares hard.
// When any row held by the wave is hard (a wave-uniform branch), the Jacobi path runs on EVERYTHING the wave holds -- for the
// packed engine that is the packed instantiation, both halves of every lane at once -- and only the hard rows keep its result.
// (Round 2 redid hard rows one lane-half at a time inside a divergent branch: fast path + two scalar SVDs per lane, and a batch
// with 1 % near-reflections took 1.9 x the time of a Gaussian one.)  signed_svd<., float> and signed_svd<., f32x2> execute the
// same IEEE operations per matrix, so a row's bits do not depend on which instantiation, or which wave-mates, it met.
// WANT_BWD: the frames come with their singular values, for the hard rows' backward (project_backward).
// (float64 rows go straight to Jacobi: so3_project_fwd_f64 is not a benchmark path.)
template <class T> struct HardRows {
    typename Tr<T>::mask hard;       // which rows took the Jacobi path (their backward must, too)
    bool any;                        // wave-uniform: does `frames` hold anything
    SignedSvd<T> frames;
    Prescale<T> prescale;            // of the rows the fast path settled (WANT_BWD only)
};
template <bool WANT_BWD, class T, bool SKIP = true> __device__ __forceinline__ void project_rotation_frames(const T (&m)[9], T (&r)[9], HardRows<T> &h) {
    typedef Tr<T> R;
    h.hard = quat_rotation<T, SKIP>(m, r, WANT_BWD ? &h.prescale : nullptr);
    h.any = R::wave_any(h.hard);
    if (__builtin_expect(h.any, 0)) {
        h.frames = signed_svd<WANT_BWD, T>(m);
        T rj[9];
        rotation_from(h.frames, rj);
#pragma unroll
        for (int j = 0; j < 9; ++j) r[j] = R::sel(h.hard, rj[j], r[j]);
    }
}
template <class T, bool SKIP = true> __device__ __forceinline__ typename Tr<T>::mask project_rotation(const T (&m)[9], T (&r)[9]) {
    HardRows<T> h;
    project_rotation_frames<false, T, SKIP>(m, r, h);
    return h.hard;
}
template <> __device__ __forceinline__ bool project_rotation<double>(const double (&m)[9], double (&r)[9]) {
    const auto f = signed_svd<false, double, 4, true, 6>(m);
    rotation_from(f, r);
    return true;
}

// sign(det M) -> flip flag.  A float32 cofactor expansion decides whenever |det| clears its own rounding bound
// (8 eps times the sum of the |terms|); only the rare rows inside that band (and rows whose products leave the
// float32 range) pay for the float64 evaluation, where products of floats are exact.
__device__ __forceinline__ bool det_negative(const float (&m)[9]) {
    const float t0 = m[4] * m[8], t1 = m[5] * m[7], t2 = m[3] * m[8], t3 = m[5] * m[6], t4 = m[3] * m[7], t5 = m[4] * m[6];
    const float det = fmaf(m[0], t0 - t1, fmaf(-m[1], t2 - t3, m[2] * (t4 - t5)));
    const float mag = fmaf(fabsf(m[0]), fabsf(t0) + fabsf(t1), fmaf(fabsf(m[1]), fabsf(t2) + fabsf(t3), fabsf(m[2]) * (fabsf(t4) + fabsf(t5))));
    if (__builtin_expect(fabsf(det) > 1e-6f * mag && mag > 1e-30f && mag < 1e30f, 1)) return det < 0.f;
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    return a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g) < 0.0;
}

// dM = U' Bm V^T for upstream G (row-major), Bm_ij = (A_ij - A_ji)/(s_i + s_j), A = U'^T G V.
#pragma clang fp contract(on)         // (see the Jacobi path)
template <class T>
__device__ __forceinline__ void project_backward(const SignedSvd<T> &f, const T (&g)[9], T (&dm)[9]) {
    typedef Tr<T> R;
    const V3<T> g0 = mk<T>(g[0], g[1], g[2]), g1 = mk<T>(g[3], g[4], g[5]), g2 = mk<T>(g[6], g[7], g[8]);
    // G v_j
    const V3<T> gv1 = mk<T>(dot(g0, f.v1), dot(g1, f.v1), dot(g2, f.v1));
    const V3<T> gv2 = mk<T>(dot(g0, f.v2), dot(g1, f.v2), dot(g2, f.v2));
    const V3<T> gv3 = mk<T>(dot(g0, f.v3), dot(g1, f.v3), dot(g2, f.v3));
    const T a12 = dot(f.u1, gv2), a21 = dot(f.u2, gv1);
    const T a13 = dot(f.u1, gv3), a31 = dot(f.u3, gv1);
    const T a23 = dot(f.u2, gv3), a32 = dot(f.u3, gv2);
    const T floor_ = R::fma(R::splat(Consts<typename R::scalar>::bwd_rel), f.s1, R::splat(Consts<typename R::scalar>::bwd_abs));
    const T k = f.inv_scale;                         // singular values are in prescaled units
    const T b12 = (a12 - a21) * k * R::rcp(R::max(f.s1 + f.s2, floor_));
    const T b13 = (a13 - a31) * k * R::rcp(R::max(f.s1 + f.s3, floor_));
    const T b23 = (a23 - a32) * k * R::rcp(R::max(f.s2 + f.s3, floor_));
    // T = U' Bm : t1 = -b12 u2 - b13 u3 ; t2 = b12 u1 - b23 u3 ; t3 = b13 u1 + b23 u2
    const V3<T> t1 = axpy<T>(-b12, f.u2, scale<T>(f.u3, -b13));
    const V3<T> t2 = axpy<T>(b12, f.u1, scale<T>(f.u3, -b23));
    const V3<T> t3 = axpy<T>(b13, f.u1, scale<T>(f.u2, b23));
    const V3<T> r0 = axpy<T>(t3.x, f.v3, axpy<T>(t2.x, f.v2, scale<T>(f.v1, t1.x)));
    const V3<T> r1 = axpy<T>(t3.y, f.v3, axpy<T>(t2.y, f.v2, scale<T>(f.v1, t1.y)));
    const V3<T> r2 = axpy<T>(t3.z, f.v3, axpy<T>(t2.z, f.v2, scale<T>(f.v1, t1.z)));
    dm[0] = r0.x; dm[1] = r0.y; dm[2] = r0.z;
    dm[3] = r1.x; dm[4] = r1.y; dm[5] = r1.z;
    dm[6] = r2.x; dm[7] = r2.y; dm[8] = r2.z;
}

#pragma clang fp contract(fast)

// =====================================================================================================================
// Backward of the projection in terms of R alone (round 2): with S = R^T M (symmetric at the optimum, eigenvalues
// s1, s2, s3') a perturbation dM turns R by R [w]x where  (tr(S) I - S) w = axial(R^T dM - dM^T R)  -- the matrix
// A = tr(S) I - S has the eigenvalues s_i + s_j, the denominators of the reference's svd_backward chain.  Transposing,
//     dL/dM = R [y]x ,   A y = axial(R^T G - G^T R) ,
// which is U' B V^T of project_backward written without U', V or the singular values (B = [y]x in V's basis).
// 18 + 18 packed instructions for the two products' needed entries, a symmetric 3x3 solve by cofactors (one v_rcp), 18
// for R [y]x: the fast path's rotation is all it needs, so K2 / K3 no longer pay for an SVD on the rows 3a settles.
// Valid where 3a's tests passed: they bound the smallest eigenvalue of A, 2 (s2 + s3'), from below.
template <class T>
__device__ __forceinline__ void backward_from_rotation(const T (&m)[9], const T (&r)[9], const T (&g)[9], T (&dm)[9]) {
    typedef Tr<T> R;
    // S = R^T M, S_ij = sum_k r[3k+i] m[3k+j], SYMMETRISED: off-diagonal entries are the mean of both triangles.  R carries a
    // round-off rotation delta ~ eps s1 / gap about its ill-conditioned axis, which makes the computed S = (I - [delta]x) S_exact
    // non-symmetric; one triangle alone then shifts A's smallest eigenvalue (s2 + s3') at FIRST order in delta -- a relative
    // error eps (s1 / gap)^2 in y -- while the symmetric part shifts it at second order only and leaves eps s1 / gap, the
    // conditioning of the gradient itself (round 2 used the upper triangle: 10-300 x the Jacobi frames' error on
    // s = (1, e, -e'), e ~ 1e-3).
    const T s00 = R::fma(r[6], m[6], R::fma(r[3], m[3], r[0] * m[0]));
    const T s11 = R::fma(r[7], m[7], R::fma(r[4], m[4], r[1] * m[1]));
    const T s22 = R::fma(r[8], m[8], R::fma(r[5], m[5], r[2] * m[2]));
    const T half = R::splat(typename R::scalar(0.5));
    const T s01 = (R::fma(r[6], m[7], R::fma(r[3], m[4], r[0] * m[1])) + R::fma(r[7], m[6], R::fma(r[4], m[3], r[1] * m[0]))) * half;
    const T s02 = (R::fma(r[6], m[8], R::fma(r[3], m[5], r[0] * m[2])) + R::fma(r[8], m[6], R::fma(r[5], m[3], r[2] * m[0]))) * half;
    const T s12 = (R::fma(r[7], m[8], R::fma(r[4], m[5], r[1] * m[2])) + R::fma(r[8], m[7], R::fma(r[5], m[4], r[2] * m[1]))) * half;
    // A = tr(S) I - S
    const T a00 = s11 + s22, a11 = s00 + s22, a22 = s00 + s11, a01 = -s01, a02 = -s02, a12 = -s12;
    // b = axial(Z - Z^T), Z = R^T G:  b = (Z21 - Z12, Z02 - Z20, Z10 - Z01)   (0-based; Z_ij = sum_k r[3k+i] g[3k+j])
    const T b0 = R::fma(r[8], g[7], R::fma(r[5], g[4], r[2] * g[1])) - R::fma(r[7], g[8], R::fma(r[4], g[5], r[1] * g[2]));
    const T b1 = R::fma(r[6], g[8], R::fma(r[3], g[5], r[0] * g[2])) - R::fma(r[8], g[6], R::fma(r[5], g[3], r[2] * g[0]));
    const T b2 = R::fma(r[7], g[6], R::fma(r[4], g[3], r[1] * g[0])) - R::fma(r[6], g[7], R::fma(r[3], g[4], r[0] * g[1]));
    // y = A^-1 b by cofactors (A symmetric positive definite on settled rows)
    const T c00 = R::fma(a11, a22, -(a12 * a12)), c01 = R::fma(a02, a12, -(a01 * a22)), c02 = R::fma(a01, a12, -(a02 * a11));
    const T c11 = R::fma(a00, a22, -(a02 * a02)), c12 = R::fma(a01, a02, -(a00 * a12)), c22 = R::fma(a00, a11, -(a01 * a01));
    const T det = R::fma(a02, c02, R::fma(a01, c01, a00 * c00));
    const T rd = R::rcp(det);
    const T y0 = R::fma(c02, b2, R::fma(c01, b1, c00 * b0)) * rd;
    const T y1 = R::fma(c12, b2, R::fma(c11, b1, c01 * b0)) * rd;
    const T y2 = R::fma(c22, b2, R::fma(c12, b1, c02 * b0)) * rd;
    // dM = R [y]x :  column 0 = y2 r_col1 - y1 r_col2, column 1 = y0 r_col2 - y2 r_col0, column 2 = y1 r_col0 - y0 r_col1
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        dm[3 * i + 0] = R::fma(y2, r[3 * i + 1], -(y1 * r[3 * i + 2]));
        dm[3 * i + 1] = R::fma(y0, r[3 * i + 2], -(y2 * r[3 * i + 0]));
        dm[3 * i + 2] = R::fma(y1, r[3 * i + 0], -(y0 * r[3 * i + 1]));
    }
}

// dM for upstream G, given the rotation R and the hard rows' frames from project_rotation_frames<true>: settled rows from the
// rotation, hard rows through the Jacobi frames (their denominators s_i + s_j may vanish: floored there).
template <class T>
__device__ __forceinline__ void backward_given_rotation(const T (&m)[9], const T (&r)[9], const T (&g)[9], const HardRows<T> &h, T (&dm)[9]) {
    typedef Tr<T> R;
    if (__builtin_expect(h.prescale.any, 0)) {
        // rows outside the fast path's scale window: S = R^T M, its cofactors and determinant are formed from the prescaled
        // matrix (third powers of the entries), and dM(c M) = dM(M) / c
        T ms[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) ms[j] = m[j] * h.prescale.factor;
        backward_from_rotation<T>(ms, r, g, dm);
#pragma unroll
        for (int j = 0; j < 9; ++j) dm[j] = dm[j] * h.prescale.factor;
    } else {
        backward_from_rotation<T>(m, r, g, dm);
    }
    if (__builtin_expect(h.any, 0)) {
        T dj[9];
        project_backward(h.frames, g, dj);
#pragma unroll
        for (int j = 0; j < 9; ++j) dm[j] = R::sel(h.hard, dj[j], dm[j]);
    }
}

// K2's arithmetic (autograd of K1): rotation, then its backward.
template <class T> __device__ __forceinline__ void project_backward_rows(const T (&m)[9], const T (&g)[9], T (&dm)[9]) {
    T r[9];
    HardRows<T> h;
    project_rotation_frames<true, T>(m, r, h);
    backward_given_rotation<T>(m, r, g, h, dm);
}
template <> __device__ __forceinline__ void project_backward_rows<double>(const double (&m)[9], const double (&g)[9], double (&dm)[9]) {
    const auto f = signed_svd<true, double, 4, true, 6>(m);
    project_backward(f, g, dm);
}

}  // namespace so3
