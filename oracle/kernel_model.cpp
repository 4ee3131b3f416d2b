// kernel_model.cpp -- the kernels' own arithmetic (poseestimation_amd/csrc/so3_device.h, compiled with SO3_HOST_MODEL)
// running on the host.  TEST INFRASTRUCTURE ONLY (oracle/): it lets the CPU test suite drive the exact templates the
// HIP kernels instantiate -- sweeps, column ordering, rank-one branch, backward -- over adversarial input without a GPU.
// Differences from the device: libm's correctly rounded sqrt / division stand in for v_rsq_f32 / v_sqrt_f32 / v_rcp_f32
// (1 ulp), and a "wave" is a single lane (the adaptive sweep is applied per matrix on the device too, so results agree
// to that rounding).  Built by oracle/kernel_model.py with clang++ (ext_vector_type is a clang extension).
#define SO3_HOST_MODEL 1
#include <stdint.h>

#include "../poseestimation_amd/csrc/so3_device.h"

extern "C" {

// one matrix per "lane" (T = float): what the tile kernels and the Kabsch kernel instantiate
void model_project_f32(const float *M, float *R, uint8_t *flip, int64_t B) {
    for (int64_t b = 0; b < B; ++b) {
        float m[9], r[9];
        for (int i = 0; i < 9; ++i) m[i] = M[9 * b + i];
        const auto f = so3::signed_svd<false, float>(m);
        so3::rotation_from(f, r);
        for (int i = 0; i < 9; ++i) R[9 * b + i] = r[i];
        if (flip) flip[b] = so3::det_negative(m) ? 1 : 0;
    }
}

// two matrices per "lane" (T = f32x2): what the streaming engine instantiates; an odd last row is paired with itself
void model_project_packed_f32(const float *M, float *R, int64_t B) {
    typedef so3::f32x2 T;
    for (int64_t b = 0; b < B; b += 2) {
        const int64_t b1 = b + 1 < B ? b + 1 : b;
        T m[9], r[9];
        for (int i = 0; i < 9; ++i) m[i] = T{M[9 * b + i], M[9 * b1 + i]};
        const auto f = so3::signed_svd<false, T>(m);
        so3::rotation_from(f, r);
        for (int i = 0; i < 9; ++i) { R[9 * b + i] = r[i].x; R[9 * b1 + i] = r[i].y; }
    }
}

void model_project_bwd_f32(const float *M, const float *G, float *dM, int64_t B) {
    for (int64_t b = 0; b < B; ++b) {
        float m[9], g[9], d[9];
        for (int i = 0; i < 9; ++i) { m[i] = M[9 * b + i]; g[i] = G[9 * b + i]; }
        const auto f = so3::signed_svd<true, float>(m);
        so3::project_backward(f, g, d);
        for (int i = 0; i < 9; ++i) dM[9 * b + i] = d[i];
    }
}

void model_project_f64(const double *M, double *R, int64_t B) {
    for (int64_t b = 0; b < B; ++b) {
        double m[9], r[9];
        for (int i = 0; i < 9; ++i) m[i] = M[9 * b + i];
        const auto f = so3::signed_svd<false, double, 4, true, 6>(m);
        so3::rotation_from(f, r);
        for (int i = 0; i < 9; ++i) R[9 * b + i] = r[i];
    }
}

void model_project_bwd_f64(const double *M, const double *G, double *dM, int64_t B) {
    for (int64_t b = 0; b < B; ++b) {
        double m[9], g[9], d[9];
        for (int i = 0; i < 9; ++i) { m[i] = M[9 * b + i]; g[i] = G[9 * b + i]; }
        const auto f = so3::signed_svd<true, double, 4, true, 6>(m);
        so3::project_backward(f, g, d);
        for (int i = 0; i < 9; ++i) dM[9 * b + i] = d[i];
    }
}

}  // extern "C"
