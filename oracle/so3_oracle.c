/* CPU oracle (plain C, float64 arithmetic) for the SVD -> SO(3) hot path.
 *
 * TEST INFRASTRUCTURE ONLY: linked/loaded by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg -- never by the product library (poseestimation_amd/csrc).
 *
 * It restates the reference formula literally --
 *     u, s, v = svd(m);  d = det(u v^T);  r = u diag(1,1,d) v^T
 *     (/root/reference/rotation_representation.py:199-205)
 * -- with its own float64 SVD, because the reference's SVD lives in a third-party dependency
 * (LAPACK ?gesdd behind torch.svd; torch==1.6.0 pinned in /root/reference/requirements.txt:7)
 * that cannot be linked from C here.  The SVD below is a textbook cyclic one-sided Jacobi with
 * explicit V accumulation, run to convergence (not the kernel's fixed-sweep, V-free variant), so
 * it is an independent second opinion.  It is pinned against the golden vectors generated from
 * the reference itself (tests/golden/, tests/test_oracle_golden.py).
 *
 * Being C it finishes 1M matrices in about a second, so the GPU parity tests use it row by row
 * at BASELINE.json's full sizes.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { double u[9], v[9], s[3]; } svd3_t;   /* row-major 3x3 */

static void svd3(const double *m, svd3_t *out)
{
    double a[3][3], v[3][3];   /* a[k] = column k of the working matrix, v[k] = column k of V */
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < 3; ++i) { a[k][i] = m[3 * i + k]; v[k][i] = (i == k); }

    static const int pairs[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int pi = 0; pi < 3; ++pi) {
            const int p = pairs[pi][0], q = pairs[pi][1];
            double al = 0, be = 0, ga = 0;
            for (int i = 0; i < 3; ++i) { al += a[p][i] * a[p][i]; be += a[q][i] * a[q][i]; ga += a[p][i] * a[q][i]; }
            if (ga == 0.0) continue;
            const double rel = fabs(ga) / sqrt(al * be);
            if (rel > off) off = rel;
            const double zeta = (be - al) / (2.0 * ga);
            const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
            for (int i = 0; i < 3; ++i) {
                const double x = a[p][i], y = a[q][i];
                a[p][i] = c * x - s * y; a[q][i] = s * x + c * y;
                const double vx = v[p][i], vy = v[q][i];
                v[p][i] = c * vx - s * vy; v[q][i] = s * vx + c * vy;
            }
        }
        if (off < 1e-16) break;
    }
    /* singular values, sorted descending (LAPACK's order, which puts the flip on the smallest) */
    double sv[3]; int ord[3] = {0, 1, 2};
    for (int k = 0; k < 3; ++k) sv[k] = sqrt(a[k][0] * a[k][0] + a[k][1] * a[k][1] + a[k][2] * a[k][2]);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (sv[ord[j]] < sv[ord[j + 1]]) { int t = ord[j]; ord[j] = ord[j + 1]; ord[j + 1] = t; }
    double u[3][3];   /* u[k] = k-th left singular vector */
    for (int k = 0; k < 3; ++k) {
        const int c = ord[k];
        out->s[k] = sv[c];
        for (int i = 0; i < 3; ++i) { out->v[3 * i + k] = v[c][i]; u[k][i] = a[c][i]; }
    }
    /* normalise; complete the basis where a singular value vanishes (rank-deficient input) */
    const double tiny = 1e-300;
    if (out->s[0] > tiny) { for (int i = 0; i < 3; ++i) u[0][i] /= out->s[0]; }
    else { u[0][0] = 1; u[0][1] = 0; u[0][2] = 0; }
    double w[3], nw;
    if (out->s[1] > tiny * 1e10 && out->s[1] > 1e-14 * out->s[0]) {
        double dot = 0; for (int i = 0; i < 3; ++i) dot += u[0][i] * u[1][i];
        for (int i = 0; i < 3; ++i) w[i] = u[1][i] - dot * u[0][i];
    } else {
        /* any unit vector orthogonal to u0: e_k x u0 with k the smallest |u0| component (z first) */
        int k = 2; if (fabs(u[0][1]) < fabs(u[0][k])) k = 1; if (fabs(u[0][0]) < fabs(u[0][k])) k = 0;
        double e[3] = {0, 0, 0}; e[k] = 1;
        w[0] = e[1] * u[0][2] - e[2] * u[0][1]; w[1] = e[2] * u[0][0] - e[0] * u[0][2]; w[2] = e[0] * u[0][1] - e[1] * u[0][0];
    }
    nw = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    for (int i = 0; i < 3; ++i) u[1][i] = w[i] / nw;
    /* third left vector: +-(u0 x u1), sign so that u2 . a_2 >= 0 (a proper SVD has s3 >= 0) */
    double x[3] = {u[0][1] * u[1][2] - u[0][2] * u[1][1], u[0][2] * u[1][0] - u[0][0] * u[1][2], u[0][0] * u[1][1] - u[0][1] * u[1][0]};
    double dot = x[0] * u[2][0] + x[1] * u[2][1] + x[2] * u[2][2];
    const double sg = (dot < 0) ? -1.0 : 1.0;
    for (int i = 0; i < 3; ++i) u[2][i] = sg * x[i];
    for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) out->u[3 * i + k] = u[k][i];
}

static double det3(const double *m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

static void matmul_abt(const double *a, const double *b, double *c)   /* c = a b^T */
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
        c[3 * i + j] = a[3 * i] * b[3 * j] + a[3 * i + 1] * b[3 * j + 1] + a[3 * i + 2] * b[3 * j + 2];
}

/* r = u diag(1,1,det(u v^T)) v^T ; returns d. */
static double project_one(const double *m, double *r, svd3_t *keep)
{
    svd3_t sv; svd3(m, &sv);
    double uvt[9]; matmul_abt(sv.u, sv.v, uvt);
    const double d = det3(uvt) < 0 ? -1.0 : 1.0;   /* det of an orthogonal matrix: +-1 */
    double vd[9]; memcpy(vd, sv.v, sizeof vd);
    vd[2] *= d; vd[5] *= d; vd[8] *= d;             /* last column of V == last row of V^T */
    matmul_abt(sv.u, vd, r);
    if (keep) *keep = sv;
    return d;
}

void oracle_project_f64(const double *M, double *R, uint8_t *flip, int64_t B)
{
    for (int64_t b = 0; b < B; ++b) {
        const double d = project_one(M + 9 * b, R + 9 * b, 0);
        if (flip) flip[b] = d < 0;
    }
}

void oracle_project_f32(const float *M, float *R, uint8_t *flip, int64_t B)
{
    for (int64_t b = 0; b < B; ++b) {
        double m[9], r[9];
        for (int i = 0; i < 9; ++i) m[i] = M[9 * b + i];
        const double d = project_one(m, r, 0);
        for (int i = 0; i < 9; ++i) R[9 * b + i] = (float)r[i];
        if (flip) flip[b] = d < 0;
    }
}

/* dM = U' B V^T, B_ij = (A_ij - A_ji)/(s'_i + s'_j), A = U'^T G V  (SURVEY section 2b, K2). */
void oracle_project_bwd_f32(const float *M, const float *G, float *dM, int64_t B)
{
    for (int64_t b = 0; b < B; ++b) {
        double m[9], g[9], r[9]; svd3_t sv;
        for (int i = 0; i < 9; ++i) { m[i] = M[9 * b + i]; g[i] = G[9 * b + i]; }
        const double d = project_one(m, r, &sv);
        double up[9]; memcpy(up, sv.u, sizeof up);
        up[2] *= d; up[5] *= d; up[8] *= d;
        const double sp[3] = {sv.s[0], sv.s[1], d * sv.s[2]};
        double gv[9], a[9], bm[9], t[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
            gv[3 * i + j] = g[3 * i] * sv.v[j] + g[3 * i + 1] * sv.v[3 + j] + g[3 * i + 2] * sv.v[6 + j];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
            a[3 * i + j] = up[i] * gv[j] + up[3 + i] * gv[3 + j] + up[6 + i] * gv[6 + j];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
            bm[3 * i + j] = (i == j) ? 0.0 : (a[3 * i + j] - a[3 * j + i]) / (sp[i] + sp[j]);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
            t[3 * i + j] = up[3 * i] * bm[j] + up[3 * i + 1] * bm[3 + j] + up[3 * i + 2] * bm[6 + j];
        double o[9]; matmul_abt(t, sv.v, o);
        for (int i = 0; i < 9; ++i) dM[9 * b + i] = (float)o[i];
    }
}

/* float64 degrees, tr(R1^T R2), clamp to [-1,1] (rotation_representation.py:230-242).
 * Returns 1 if any cos is outside [-1.1, 1.1] (where the reference raises ValueError). */
int oracle_angle_error(const float *R1, const float *R2, double *deg, int64_t B)
{
    int bad = 0;
    for (int64_t b = 0; b < B; ++b) {
        double tr = 0;
        for (int i = 0; i < 9; ++i) tr += (double)R1[9 * b + i] * (double)R2[9 * b + i];
        double c = (tr - 1.0) / 2.0;
        if (c < -1.1 || c > 1.1) bad = 1;
        if (c > 1.0) c = 1.0;
        if (c < -1.0) c = -1.0;
        deg[b] = acos(c) * (180.0 / 3.14159265358979323846);
    }
    return bad;
}

/* H_b = sum_i q_i p_i^T (float64 accumulation), R_b = proj(H_b).  P, Q: (B, N, 3) float32. */
void oracle_kabsch_f32(const float *P, const float *Q, float *R, double *Hout, int64_t B, int32_t N)
{
    for (int64_t b = 0; b < B; ++b) {
        double h[9] = {0}, r[9];
        const float *p = P + (size_t)b * N * 3, *q = Q + (size_t)b * N * 3;
        for (int32_t i = 0; i < N; ++i)
            for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c)
                h[3 * a + c] += (double)q[3 * i + a] * (double)p[3 * i + c];
        project_one(h, r, 0);
        for (int i = 0; i < 9; ++i) R[9 * b + i] = (float)r[i];
        if (Hout) for (int i = 0; i < 9; ++i) Hout[9 * b + i] = h[i];
    }
}
