// kernel_model.cpp -- the kernels' own arithmetic (poseestimation_amd/csrc/so3_device.h, compiled with SO3_HOST_MODEL)
// running on the host.  TEST INFRASTRUCTURE ONLY (oracle/): it lets the CPU test suite drive the exact templates the
// HIP kernels instantiate -- sweeps, column ordering, rank-one branch, backward -- over adversarial input without a GPU.
// Differences from the device: libm's correctly rounded sqrt / division stand in for v_rsq_f32 / v_sqrt_f32 / v_rcp_f32
// (1 ulp), and a "wave" is a single lane (the adaptive sweep is applied per matrix on the device too, so results agree
// to that rounding).  Built by oracle/kernel_model.py with clang++ (ext_vector_type is a clang extension).
#define SO3_HOST_MODEL 1
#include <stdint.h>

#include "../poseestimation_amd/csrc/so3_rows.h"      // includes so3_device.h; engine and K1..K4 store paths are device-only

extern "C" {

// one matrix per "lane" (T = float): what the tile kernels and the Kabsch kernel instantiate
void model_project_f32(const float *M, float *R, uint8_t *flip, int64_t B) {
    for (int64_t b = 0; b < B; ++b) {
        float m[9], r[9];
        for (int i = 0; i < 9; ++i) m[i] = M[9 * b + i];
        so3::project_rotation<float>(m, r);
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
        so3::project_rotation<T>(m, r);
        for (int i = 0; i < 9; ++i) { R[9 * b + i] = r[i].x; R[9 * b1 + i] = r[i].y; }
    }
}

// the Jacobi path alone (what hard rows and the backward kernels run)
void model_project_jacobi_f32(const float *M, float *R, int64_t B) {
    for (int64_t b = 0; b < B; ++b) {
        float m[9], r[9];
        for (int i = 0; i < 9; ++i) m[i] = M[9 * b + i];
        const auto f = so3::signed_svd<false, float>(m);
        so3::rotation_from(f, r);
        for (int i = 0; i < 9; ++i) R[9 * b + i] = r[i];
    }
}

// the fast path alone: R as it leaves quat_rotation and the hard flag (hard rows' R is whatever the path produced)
void model_project_quat_f32(const float *M, float *R, uint8_t *hard, int64_t B) {
    for (int64_t b = 0; b < B; ++b) {
        float m[9], r[9];
        for (int i = 0; i < 9; ++i) m[i] = M[9 * b + i];
        hard[b] = so3::quat_rotation<float>(m, r) ? 1 : 0;
        for (int i = 0; i < 9; ++i) R[9 * b + i] = r[i];
    }
}

void model_project_bwd_f32(const float *M, const float *G, float *dM, int64_t B) {
    for (int64_t b = 0; b < B; ++b) {
        float m[9], g[9], d[9];
        for (int i = 0; i < 9; ++i) { m[i] = M[9 * b + i]; g[i] = G[9 * b + i]; }
        so3::project_backward_rows<float>(m, g, d);
        for (int i = 0; i < 9; ++i) dM[9 * b + i] = d[i];
    }
}

// the Jacobi frames' backward alone (what hard rows run)
void model_project_bwd_jacobi_f32(const float *M, const float *G, float *dM, int64_t B) {
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

// ---- the pure per-row operations of so3_rows.h (heads f2 / f5, SE(3) update f1), one row at a time ----------------
namespace {
template <class Op> void run_rows(Op op, const float *in0, const float *in1, const float *in2, float *out, int64_t B) {
    for (int64_t b = 0; b < B; ++b) {
        so3::Rows<float, Op> rows;
        so3::RowCtx<1> ctx{};
        for (int i = 0; i < Op::kIn0N; ++i) rows.a[i] = in0[b * Op::kIn0N + i];
        if (Op::kIn1 != 0) for (int i = 0; i < Op::kIn1N; ++i) rows.b[i] = in1[b * Op::kIn1N + i];
        if (Op::kIn2 != 0) for (int i = 0; i < Op::kIn2N; ++i) rows.c[i] = in2[b * Op::kIn2N + i];
        op.template compute<float, 1>(rows, ctx);
        for (int i = 0; i < Op::kOut0N; ++i) out[b * Op::kOut0N + i] = rows.o0[i];
    }
}
}  // namespace

extern "C" {
#define MODEL_HEAD(NAME, FWD, BWD)                                                                             \
    void model_##NAME##_fwd(const float *X, float *R, int64_t B) { run_rows(FWD{}, X, nullptr, nullptr, R, B); } \
    void model_##NAME##_bwd(const float *X, const float *G, float *dX, int64_t B) { run_rows(BWD{}, X, G, nullptr, dX, B); }
MODEL_HEAD(quat, so3::OpQuat<false>, so3::OpQuat<true>)
MODEL_HEAD(euler, so3::OpEuler<false>, so3::OpEuler<true>)
MODEL_HEAD(ortho5d, so3::OpOrtho5d<false>, so3::OpOrtho5d<true>)
MODEL_HEAD(expmap, so3::OpExpMap<false>, so3::OpExpMap<true>)
MODEL_HEAD(ortho6d, so3::OpOrtho6d, so3::OpOrtho6dBwd)
#undef MODEL_HEAD

void model_se3_update(const float *out12, const float *Tinit, float *Tpred, float fx, float fy, int64_t B) {
    so3::OpSe3Update op; op.inv_fx = 1.0f / fx; op.inv_fy = 1.0f / fy;
    run_rows(op, out12, Tinit, nullptr, Tpred, B);
}
void model_se3_update_bwd(const float *out12, const float *Tinit, const float *G, float *dout12, float fx, float fy, int64_t B) {
    so3::OpSe3UpdateBwd op; op.inv_fx = 1.0f / fx; op.inv_fy = 1.0f / fy;
    run_rows(op, out12, Tinit, G, dout12, B);
}
}  // extern "C"

// The parked-rows list's reservation protocol (so3_rows.h: park_reserve_protocol) under a replayed interleaving: actor 0 reads the
// count, then actors 1..k-1 run their whole reservations (in order; actor j > 0 itself is interrupted by actor j+1.. when `nested`),
// then actor 0's compare-and-swap is attempted -- and retried, now undisturbed, if it finds the count changed.  base[j] = the entry
// actor j was given or -1.  Returns the final count.  (Every side effect of the protocol is one successful compare-and-swap, so
// any schedule of the device's waves is equivalent to some such nesting.)
namespace {
struct ParkReplay {
    unsigned *count; const unsigned *n; int *base; int k; unsigned cap; bool nested;
    void run(int j) {
        bool disturbed = false;
        base[j] = so3::park_reserve_protocol(
            count, n[j], cap,
            [&](unsigned *p) {
                const unsigned v = *p;
                if (!disturbed) {
                    disturbed = true;
                    if (j == 0 && !nested) { for (int i = 1; i < k; ++i) run(i); }
                    else if (nested && j + 1 < k) run(j + 1);
                }
                return v;
            },
            [&](unsigned *p, unsigned expected, unsigned desired) {
                const unsigned seen = *p;
                if (seen == expected) *p = desired;
                return seen;
            });
    }
};
}  // namespace

extern "C" unsigned model_park_reserve_interleaved(unsigned start, unsigned cap, const unsigned *n, int k, int nested, int *base) {
    unsigned count = start;
    ParkReplay r{&count, n, base, k, cap, nested != 0};
    r.run(0);
    return count;
}

// the fast path's own statistics since the last call (so3_device.h: HostCounters)
extern "C" void model_fast_path_counters(long long *out2, int reset) {
    out2[0] = so3::host_counters().refined_rows;
    out2[1] = so3::host_counters().refinements;
    if (reset) so3::host_counters() = so3::HostCounters{};
}
