// so3proj.hip -- C ABI of libso3proj.so (gfx950 only) and the small-batch kernels.  See include/so3proj.h.
//
// Large, 16-byte-aligned batches run on the row-streaming engine of so3_rows.h (persistent waves, packed
// two-matrices-per-lane arithmetic, branch-free buffer I/O).  What is left -- a remainder of < 64 rows,
// pointers that are not 16-byte aligned, and the Kabsch kernel -- lives here: 256-thread workgroups own a
// contiguous tile of 256 rows (9216 B), move it with coalesced accesses through LDS, and each lane picks its
// nine floats out of LDS at a 9-dword stride (odd stride -> conflict-free ds_read_b32 / ds_write_b32).
#include <hip/hip_runtime.h>
#include <cxxabi.h>
#include <stdlib.h>
#include <string>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <initializer_list>

#include "../../include/so3proj.h"
#include "so3_device.h"
#include "so3_rows.h"

namespace {



constexpr int kBlock = 256;                 // lanes (= 3x3 blocks) per workgroup tile
constexpr int kTileFloats = kBlock * 9;     // 2304 floats = 9216 B
constexpr int kTileVec4 = kTileFloats / 4;  // 576 float4

thread_local char g_err[256] = "";

int fail(int code, const char *what) {
    snprintf(g_err, sizeof g_err, "%s: %s", what, code == SO3_ERR_INVALID ? "invalid argument" : hipGetErrorString((hipError_t)code));
    return code;
}

int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail((int)e, what);
}

__host__ __device__ inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- tile movers ----------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(static_cast<uint32_t>(b) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    const __bf16 h = static_cast<__bf16>(f);   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    uint16_t u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}

// Global (f32) -> LDS tile.  `n` = valid blocks in this tile (<= 256).  VEC: 16-byte path allowed.
template <bool VEC>
__device__ __forceinline__ void tile_in_f32(const float *__restrict__ g, int n, float *tile) {
    const int tid = threadIdx.x;
    if (VEC && n == kBlock) {
        const float4 *src = reinterpret_cast<const float4 *>(g);
        float4 *dst = reinterpret_cast<float4 *>(tile);
        const float4 a = src[tid], b = src[tid + kBlock];
        float4 c;
        if (tid < kTileVec4 - 2 * kBlock) c = src[tid + 2 * kBlock];
        dst[tid] = a;
        dst[tid + kBlock] = b;
        if (tid < kTileVec4 - 2 * kBlock) dst[tid + 2 * kBlock] = c;
    } else {
        for (int i = tid; i < n * 9; i += kBlock) tile[i] = g[i];
    }
}

template <bool VEC>
__device__ __forceinline__ void tile_out_f32(float *__restrict__ g, int n, const float *tile) {
    const int tid = threadIdx.x;
    if (VEC && n == kBlock) {
        float4 *dst = reinterpret_cast<float4 *>(g);
        const float4 *src = reinterpret_cast<const float4 *>(tile);
        dst[tid] = src[tid];
        dst[tid + kBlock] = src[tid + kBlock];
        if (tid < kTileVec4 - 2 * kBlock) dst[tid + 2 * kBlock] = src[tid + 2 * kBlock];
    } else {
        for (int i = tid; i < n * 9; i += kBlock) g[i] = tile[i];
    }
}

// bf16 in global <-> f32 in the LDS tile (conversion on the way through).
__device__ __forceinline__ void tile_in_bf16(const uint16_t *__restrict__ g, int n, float *tile) {
    for (int i = threadIdx.x; i < n * 9; i += kBlock) tile[i] = bf16_to_f32(g[i]);
}
__device__ __forceinline__ void tile_out_bf16(uint16_t *__restrict__ g, int n, const float *tile) {
    for (int i = threadIdx.x; i < n * 9; i += kBlock) g[i] = f32_to_bf16(tile[i]);
}

template <bool BF16, bool VEC>
__device__ __forceinline__ void tile_in(const void *base, int64_t first, int n, float *tile) {
    if (BF16) tile_in_bf16(static_cast<const uint16_t *>(base) + first * 9, n, tile);
    else tile_in_f32<VEC>(static_cast<const float *>(base) + first * 9, n, tile);
}
template <bool BF16, bool VEC>
__device__ __forceinline__ void tile_out(void *base, int64_t first, int n, const float *tile) {
    if (BF16) tile_out_bf16(static_cast<uint16_t *>(base) + first * 9, n, tile);
    else tile_out_f32<VEC>(static_cast<float *>(base) + first * 9, n, tile);
}

__device__ __forceinline__ void lane_get(const float *tile, bool active, float (&m)[9]) {
    const float *p = tile + threadIdx.x * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = active ? p[i] : ((i & 3) == 0 ? 1.f : 0.f);   // idle lanes: identity
}
__device__ __forceinline__ void lane_put(float *tile, const float (&m)[9]) {
    float *p = tile + threadIdx.x * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) p[i] = m[i];
}

// ---- K1 -------------------------------------------------------------------------------------------
template <bool BF16, bool VEC, bool FLIP>
__global__ __launch_bounds__(kBlock) void k_project_fwd(const void *__restrict__ M, float *__restrict__ R,
                                                        uint8_t *__restrict__ flip, int64_t B) {
    __shared__ __attribute__((aligned(16))) float tile[kTileFloats];
    const int64_t first = static_cast<int64_t>(blockIdx.x) * kBlock;
    const int n = static_cast<int>(min<int64_t>(kBlock, B - first));
    tile_in<BF16, VEC>(M, first, n, tile);
    __syncthreads();
    float m[9], r[9];
    const bool active = static_cast<int>(threadIdx.x) < n;
    lane_get(tile, active, m);
    so3::project_rotation<float>(m, r);
    if (FLIP && active) flip[first + threadIdx.x] = so3::det_negative(m) ? 1 : 0;
    lane_put(tile, r);          // each lane overwrites only the nine words it alone has read
    __syncthreads();
    tile_out<false, VEC>(R, first, n, tile);
}


// ---- K2 -------------------------------------------------------------------------------------------
template <bool BF16, bool VEC>
__global__ __launch_bounds__(kBlock) void k_project_bwd(const void *__restrict__ M, const float *__restrict__ G,
                                                        void *__restrict__ dM, int64_t B) {
    __shared__ __attribute__((aligned(16))) float tile_m[kTileFloats];
    __shared__ __attribute__((aligned(16))) float tile_g[kTileFloats];
    const int64_t first = static_cast<int64_t>(blockIdx.x) * kBlock;
    const int n = static_cast<int>(min<int64_t>(kBlock, B - first));
    tile_in<BF16, VEC>(M, first, n, tile_m);
    tile_in<false, VEC>(G, first, n, tile_g);
    __syncthreads();
    float m[9], g[9], dm[9];
    const bool active = static_cast<int>(threadIdx.x) < n;
    lane_get(tile_m, active, m);
    lane_get(tile_g, active, g);
    so3::project_backward_rows<float>(m, g, dm);
    lane_put(tile_m, dm);
    __syncthreads();
    tile_out<BF16, VEC>(dM, first, n, tile_m);
}

// ---- block reduction of one double (sum) --------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double block_sum(double v, double *scratch /* >= 4 doubles of LDS */) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) scratch[wave] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

// A remainder kernel's share of a workspace reduction (so3_rows.h): its workgroups fill slots [0, gridDim.x) and take no
// ticket; the engine launch that follows on the stream sums them with its own.
__device__ __forceinline__ void publish_partial(so3::ReduceWs *ws, double total, bool flag) {
    so3::slot_publish(ws, blockIdx.x, total);
    if (flag) atomicOr(&ws->flag, 1);
}

// ---- K3 -------------------------------------------------------------------------------------------
template <bool BF16, bool VEC, bool WANT_R, bool WANT_DM>
__global__ __launch_bounds__(kBlock) void k_frob_fwd_bwd(const void *__restrict__ M, const float *__restrict__ Rtrue,
                                                         float *__restrict__ R, void *__restrict__ dM,
                                                         double *__restrict__ loss_sum, int64_t B, float inv_b, so3::ReduceWs *ws) {
    __shared__ __attribute__((aligned(16))) float tile_m[kTileFloats];
    __shared__ __attribute__((aligned(16))) float tile_t[kTileFloats];
    __shared__ double red[4];
    const int64_t ntiles = (B + kBlock - 1) / kBlock;
    double acc = 0.0;                                  // per-lane partial of sum_b ||.||_F over this workgroup's tiles
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t first = tile * kBlock;
        const int n = static_cast<int>(min<int64_t>(kBlock, B - first));
        tile_in<BF16, VEC>(M, first, n, tile_m);
        tile_in<false, VEC>(Rtrue, first, n, tile_t);
        __syncthreads();
        float m[9], t[9], r[9], g[9], dm[9];
        const bool active = static_cast<int>(threadIdx.x) < n;
        lane_get(tile_m, active, m);
        lane_get(tile_t, active, t);
        so3::HardRows<float> hard;
        so3::project_rotation_frames<WANT_DM, float>(m, r, hard);
        float n2 = 0.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            g[i] = r[i] - t[i];                       // d||Rtrue - R||/dR = (R - Rtrue)/||.||
            n2 = fmaf(g[i], g[i], n2);
        }
        const float nrm = n2 * __builtin_amdgcn_rsqf(fmaxf(n2, 1e-37f));
        const float gs = (n2 > 0.f) ? inv_b * __builtin_amdgcn_rsqf(n2) : 0.f;    // zero difference -> zero gradient
        if (active) acc += static_cast<double>(nrm);
        if (WANT_DM) {
#pragma unroll
            for (int i = 0; i < 9; ++i) g[i] *= gs;
            so3::backward_given_rotation<float>(m, r, g, hard, dm);
            lane_put(tile_m, dm);
        }
        if (WANT_R) lane_put(tile_t, r);
        __syncthreads();
        if (WANT_DM) tile_out<BF16, VEC>(dM, first, n, tile_m);
        if (WANT_R) tile_out<false, VEC>(R, first, n, tile_t);
        __syncthreads();                               // the tiles are reused by the next iteration
    }
    // one float64 atomic per workgroup: same-address atomics cost ~12 ns each (one per 256-row tile made this
    // kernel atomic-bound: 59 us per 1M rows)
    const double total = block_sum(acc, red);
    if (threadIdx.x == 0) {
        if (ws != nullptr) publish_partial(ws, total, false);
        else atomicAdd(loss_sum, total);
    }
}

// K3 for a batch that fits ONE workgroup (config #4: B = 512): one row per thread, rows read and written straight from
// global memory (9 KB in all: latency-bound, not a streaming problem), the loss reduced inside the workgroup and written
// with a plain store -- no zero-fill launch before the kernel and no atomic.  The call is then a single launch.
constexpr int kSmallBatch = 1024;
template <bool BF16, bool WANT_R, bool WANT_DM>
__global__ __launch_bounds__(kSmallBatch) void k_frob_small(const void *__restrict__ M, const float *__restrict__ Rtrue,
                                                            float *__restrict__ R, void *__restrict__ dM,
                                                            double *__restrict__ loss_sum, float *__restrict__ loss_mean, int B, float inv_b) {
    __shared__ double red[kSmallBatch / 64];
    const int b = threadIdx.x;
    const bool active = b < B;
    float m[9], t[9], r[9], g[9], dm[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        m[i] = (i & 3) == 0 ? 1.f : 0.f;              // idle lanes: identity
        t[i] = m[i];
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            m[i] = BF16 ? bf16_to_f32(static_cast<const uint16_t *>(M)[b * 9 + i]) : static_cast<const float *>(M)[b * 9 + i];
            t[i] = Rtrue[b * 9 + i];
        }
    }
    so3::HardRows<float> hard;
    so3::project_rotation_frames<WANT_DM, float>(m, r, hard);
    float n2 = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        g[i] = r[i] - t[i];
        n2 = fmaf(g[i], g[i], n2);
    }
    const float nrm = n2 * __builtin_amdgcn_rsqf(fmaxf(n2, 1e-37f));
    const float gs = (n2 > 0.f) ? inv_b * __builtin_amdgcn_rsqf(n2) : 0.f;        // zero difference -> zero gradient
    if (WANT_DM) {
#pragma unroll
        for (int i = 0; i < 9; ++i) g[i] *= gs;
        so3::backward_given_rotation<float>(m, r, g, hard, dm);
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            if (WANT_DM) {
                if (BF16) static_cast<uint16_t *>(dM)[b * 9 + i] = f32_to_bf16(dm[i]);
                else static_cast<float *>(dM)[b * 9 + i] = dm[i];
            }
            if (WANT_R) R[b * 9 + i] = r[i];
        }
    }
    const double v = wave_sum(active ? static_cast<double>(nrm) : 0.0);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double total = 0.0;
        for (int w = 0; w < static_cast<int>(blockDim.x >> 6); ++w) total += red[w];
        if (loss_sum != nullptr) *loss_sum = total;
        if (loss_mean != nullptr) *loss_mean = static_cast<float>(total * (1.0 / static_cast<double>(B)));   // float64 sum / B, rounded once
    }
}

// ---- K3', stand-alone Frobenius loss (3D-Pose/loss.py:7-11) for callers that already hold R_pred -----
// loss_sum += sum_b ||Rtrue_b - Rpred_b||_F ;  optional dRpred_b = (Rpred_b - Rtrue_b) / (B ||.||_F).
// Element-wise over flat float4s is not possible (the norm is per 9-float row), so the same 256-row tile
// staging as the other block kernels is used.
template <bool VEC, bool WANT_GRAD>
__global__ __launch_bounds__(kBlock) void k_frob_loss(const float *__restrict__ Rpred, const float *__restrict__ Rtrue,
                                                      float *__restrict__ dRpred, double *__restrict__ loss_sum,
                                                      int64_t B, float inv_b, so3::ReduceWs *ws) {
    __shared__ __attribute__((aligned(16))) float tile_p[kTileFloats];
    __shared__ __attribute__((aligned(16))) float tile_t[kTileFloats];
    __shared__ double red[4];
    const int64_t ntiles = (B + kBlock - 1) / kBlock;
    double acc = 0.0;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t first = tile * kBlock;
        const int n = static_cast<int>(min<int64_t>(kBlock, B - first));
        tile_in<false, VEC>(Rpred, first, n, tile_p);
        tile_in<false, VEC>(Rtrue, first, n, tile_t);
        __syncthreads();
        float p[9], t[9], g[9];
        const bool active = static_cast<int>(threadIdx.x) < n;
        lane_get(tile_p, active, p);
        lane_get(tile_t, active, t);
        float n2 = 0.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) { g[i] = p[i] - t[i]; n2 = fmaf(g[i], g[i], n2); }
        const float nrm = n2 * __builtin_amdgcn_rsqf(fmaxf(n2, 1e-37f));
        if (active) acc += static_cast<double>(nrm);
        __syncthreads();
        if (WANT_GRAD) {
            const float gs = (n2 > 0.f) ? inv_b * __builtin_amdgcn_rsqf(n2) : 0.f;     // zero difference -> zero gradient
#pragma unroll
            for (int i = 0; i < 9; ++i) g[i] *= gs;
            lane_put(tile_p, g);
            __syncthreads();
            tile_out<false, VEC>(dRpred, first, n, tile_p);
            __syncthreads();
        }
    }
    const double total = block_sum(acc, red);          // one atomic per workgroup
    if (threadIdx.x == 0) {
        if (ws != nullptr) publish_partial(ws, total, false);
        else atomicAdd(loss_sum, total);
    }
}

// K3' for a launch-bound batch (B <= kSmallBatch): one workgroup, one row per thread, loss_sum written with a plain
// store -- no zero-fill before the kernel.  Row arithmetic as in k_frob_loss / OpFrobLoss.
template <bool WANT_GRAD>
__global__ __launch_bounds__(kSmallBatch) void k_frob_loss_small(const float *__restrict__ Rpred, const float *__restrict__ Rtrue,
                                                                 float *__restrict__ dRpred, double *__restrict__ loss_sum,
                                                                 float *__restrict__ loss_mean, int B, float inv_b) {
    __shared__ double red[kSmallBatch / 64];
    const int b = threadIdx.x;
    const bool active = b < B;
    float g[9];
    float n2 = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        g[i] = active ? Rpred[b * 9 + i] - Rtrue[b * 9 + i] : 0.f;
        n2 = fmaf(g[i], g[i], n2);
    }
    const float nrm = n2 * __builtin_amdgcn_rsqf(fmaxf(n2, 1e-37f));
    if (WANT_GRAD && active) {
        const float gs = (n2 > 0.f) ? inv_b * __builtin_amdgcn_rsqf(n2) : 0.f;     // zero difference -> zero gradient
#pragma unroll
        for (int i = 0; i < 9; ++i) dRpred[b * 9 + i] = g[i] * gs;
    }
    const double v = wave_sum(active ? static_cast<double>(nrm) : 0.0);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double total = 0.0;
        for (int w = 0; w < static_cast<int>(blockDim.x >> 6); ++w) total += red[w];
        *loss_sum = total;
        if (loss_mean != nullptr) *loss_mean = static_cast<float>(total * (1.0 / static_cast<double>(B)));
    }
}

// dst[i] = src[i] * (*factor): the chain rule's last step for a gradient that was computed at unit upstream scale in the forward
// launch (K3 stores d loss / dM; autograd hands loss.backward() the upstream factor as a 0-dim DEVICE tensor).  One launch
// instead of torch's float() / mul / to(bfloat16) chain around a 4-us kernel.  16 bytes per thread; n counts elements.
template <bool BF16>
__global__ __launch_bounds__(kBlock) void k_scale(const void *__restrict__ src, const float *__restrict__ factor, void *__restrict__ dst, int64_t n) {
    const float f = *factor;
    constexpr int kPer = BF16 ? 8 : 4;
    const int64_t i0 = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * kPer;
    if (i0 + kPer <= n && aligned16(static_cast<const char *>(src) + i0 * (BF16 ? 2 : 4)) && aligned16(static_cast<char *>(dst) + i0 * (BF16 ? 2 : 4))) {
        if constexpr (BF16) {
            const uint4 v = *reinterpret_cast<const uint4 *>(static_cast<const uint16_t *>(src) + i0);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            uint32_t o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] = static_cast<uint32_t>(f32_to_bf16(bf16_to_f32(static_cast<uint16_t>(w[k] & 0xFFFFu)) * f))
                       | static_cast<uint32_t>(f32_to_bf16(bf16_to_f32(static_cast<uint16_t>(w[k] >> 16)) * f)) << 16;
            *reinterpret_cast<uint4 *>(static_cast<uint16_t *>(dst) + i0) = uint4{o[0], o[1], o[2], o[3]};
        } else {
            float4 v = *reinterpret_cast<const float4 *>(static_cast<const float *>(src) + i0);
            v.x *= f; v.y *= f; v.z *= f; v.w *= f;
            *reinterpret_cast<float4 *>(static_cast<float *>(dst) + i0) = v;
        }
    } else {
        for (int64_t i = i0; i < n && i < i0 + kPer; ++i) {
            if constexpr (BF16) static_cast<uint16_t *>(dst)[i] = f32_to_bf16(bf16_to_f32(static_cast<const uint16_t *>(src)[i]) * f);
            else static_cast<float *>(dst)[i] = static_cast<const float *>(src)[i] * f;
        }
    }
}

// float32 mean from the float64 sum once the kernels before it on the stream are done (batches too large for one workgroup,
// no workspace: the atomics' total is only complete at the end of the launch)
__global__ void k_mean_from_sum(const double *__restrict__ loss_sum, float *__restrict__ loss_mean, double inv_b) {
    *loss_mean = static_cast<float>(*loss_sum * inv_b);
}

// ---- K4 -------------------------------------------------------------------------------------------
template <bool VEC, bool WANT_DEG, bool WANT_SUM>
__global__ __launch_bounds__(kBlock) void k_angle_error(const float *__restrict__ R1, const float *__restrict__ R2,
                                                        double *__restrict__ out, double *__restrict__ sum_count,
                                                        int32_t *__restrict__ range_flag, double unit, int64_t B, so3::ReduceWs *ws) {
    __shared__ __attribute__((aligned(16))) float tile_a[kTileFloats];
    __shared__ __attribute__((aligned(16))) float tile_b[kTileFloats];
    __shared__ double red[4];
    const int64_t first = static_cast<int64_t>(blockIdx.x) * kBlock;
    const int n = static_cast<int>(min<int64_t>(kBlock, B - first));
    tile_in<false, VEC>(R1, first, n, tile_a);
    tile_in<false, VEC>(R2, first, n, tile_b);
    __syncthreads();
    float a[9], b[9];
    const bool active = static_cast<int>(threadIdx.x) < n;
    lane_get(tile_a, active, a);
    lane_get(tile_b, active, b);
    double tr = 0.0;                                   // tr(R1^T R2) = sum_ij R1_ij R2_ij, float64
#pragma unroll
    for (int i = 0; i < 9; ++i) tr = fma(static_cast<double>(a[i]), static_cast<double>(b[i]), tr);
    const double c_raw = (tr - 1.0) * 0.5;
    const bool bad = active && (c_raw < -1.1 || c_raw > 1.1);  // NaN compares false, as torch.any(...) does
    double c = fmin(fmax(c_raw, -1.0), 1.0);                   // torch.clamp ...
    if (c_raw != c_raw) c = c_raw;                             // ... which keeps NaN (fmin/fmax drop it)
    const double ang = so3::acos_f64(c) * unit;
    if (__any(bad) && (threadIdx.x & 63) == 0) {
        if (ws != nullptr) atomicOr(&ws->flag, 1);
        else if (range_flag != nullptr) atomicOr(range_flag, 1);
    }
    if (WANT_DEG && active) out[first + threadIdx.x] = ang;
    if (WANT_SUM) {
        const double total = block_sum(active ? ang : 0.0, red);
        if (threadIdx.x == 0) {
            if (ws != nullptr) publish_partial(ws, total, false);
            else atomicAdd(sum_count, total);  // the row count is written once by k_angle_init
        }
    }
}

// Batches of up to kSmallBatch rows (the per-batch metric of a training loop, 3D-Pose/main.py:60-62: B = 512) are
// launch-latency-bound, and the accumulators' initialisation was a launch of its own.  One workgroup, one row per thread:
// the kernel writes (sum, count) and the range flag with plain stores -- one launch, no atomics.  PROJECT: the first
// operand is the head's input M and the angle is taken of its projection (the fused evaluation step).
template <bool PROJECT, bool WANT_R, bool WANT_DEG>
__global__ __launch_bounds__(kSmallBatch) void k_angle_small(const float *__restrict__ A, const float *__restrict__ Bm,
                                                             float *__restrict__ R, double *__restrict__ deg,
                                                             double *__restrict__ sum_count, int32_t *__restrict__ range_flag,
                                                             double unit, int B) {
    __shared__ double red[kSmallBatch / 64];
    __shared__ int red_bad[kSmallBatch / 64];
    const int b = threadIdx.x;
    const bool active = b < B;
    float a[9], t[9], r[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        a[i] = (i & 3) == 0 ? 1.f : 0.f;              // idle lanes: identity
        t[i] = a[i];
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            a[i] = A[b * 9 + i];
            t[i] = Bm[b * 9 + i];
        }
    }
    if (PROJECT) {
        so3::project_rotation<float>(a, r);
        if (WANT_R && active) {
#pragma unroll
            for (int i = 0; i < 9; ++i) R[b * 9 + i] = r[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) r[i] = a[i];
    }
    double tr = 0.0;                                   // the arithmetic of OpAngle / k_angle_error, operation for operation
#pragma unroll
    for (int i = 0; i < 9; ++i) tr = fma(static_cast<double>(r[i]), static_cast<double>(t[i]), tr);
    const double c_raw = (tr - 1.0) * 0.5;
    const bool bad = active && (c_raw < -1.1 || c_raw > 1.1);
    double c = fmin(fmax(c_raw, -1.0), 1.0);
    if (c_raw != c_raw) c = c_raw;
    const double ang = so3::acos_f64(c) * unit;
    if (WANT_DEG && active) deg[b] = ang;
    const double v = wave_sum(active ? ang : 0.0);
    const bool any_bad = __any(bad);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = v; red_bad[threadIdx.x >> 6] = any_bad ? 1 : 0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double total = 0.0;
        int flag = 0;
        for (int w = 0; w < static_cast<int>(blockDim.x >> 6); ++w) { total += red[w]; flag |= red_bad[w]; }
        if (sum_count) { sum_count[0] = total; sum_count[1] = static_cast<double>(B); }
        if (range_flag) *range_flag = flag;
    }
}

__global__ void k_angle_init(double *sum_count, int32_t *range_flag, double count) {
    if (sum_count) { sum_count[0] = 0.0; sum_count[1] = count; }
    if (range_flag) *range_flag = 0;
}

// float32 radians variant (rotation_representation.py:209-227; point_cloud/main.py:61-73): tr(m1 m2^T) in float32, clamp to [lo, hi];
// `sum` (nullable): the block's angles are added to it (one atomic per workgroup); theta nullable then.
template <bool VEC>
__global__ __launch_bounds__(kBlock) void k_geodesic_f32(const float *__restrict__ R1, const float *__restrict__ R2,
                                                         float *__restrict__ theta, double *__restrict__ sum, float lo, float hi, int64_t B,
                                                         so3::ReduceWs *__restrict__ ws) {
    __shared__ __attribute__((aligned(16))) float tile_a[kTileFloats];
    __shared__ __attribute__((aligned(16))) float tile_b[kTileFloats];
    __shared__ double part[kBlock / 64];
    const int64_t first = static_cast<int64_t>(blockIdx.x) * kBlock;
    const int n = static_cast<int>(min<int64_t>(kBlock, B - first));
    tile_in<false, VEC>(R1, first, n, tile_a);
    tile_in<false, VEC>(R2, first, n, tile_b);
    __syncthreads();
    float a[9], b[9];
    const bool active = static_cast<int>(threadIdx.x) < n;
    lane_get(tile_a, active, a);
    lane_get(tile_b, active, b);
    // diagonal of m1 m2^T, summed in the reference's order: m00 + m11 + m22
    const float d0 = fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]));
    const float d1 = fmaf(a[5], b[5], fmaf(a[4], b[4], a[3] * b[3]));
    const float d2 = fmaf(a[8], b[8], fmaf(a[7], b[7], a[6] * b[6]));
    float c = (d0 + d1 + d2 - 1.f) * 0.5f;
    c = (c > hi) ? hi : c;     // torch.min / torch.max with a constant, torch.clamp: NaN stays NaN
    c = (c < lo) ? lo : c;
    const float th = acosf(c);
    if (active && theta != nullptr) theta[first + threadIdx.x] = th;
    if (sum != nullptr) {
        double v = active ? static_cast<double>(th) : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < kBlock / 64; ++w) t += part[w];
            if (ws != nullptr) publish_partial(ws, t, false);
            else atomicAdd(sum, t);
        }
    }
}

// ---- K4b: backward of the metrics, one row per thread: the remainder (< 64 rows) and unaligned views of the streaming kernel; the
// arithmetic is the engine operation's own (so3::OpAngleBwd::compute, one matrix per lane).  `g` = the per-row upstream gradient
// (float32, or float64 with F64MATH) when the operation takes one.
template <class Op>
__global__ __launch_bounds__(kBlock) void k_angle_bwd_rows(Op op, const void *__restrict__ g, float *__restrict__ d1, float *__restrict__ d2, int64_t B) {
    const int64_t row = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (row >= B) return;
    so3::RowCtx<1> ctx{};
    so3::Rows<float, Op> rows;
    const float *a = static_cast<const float *>(op.in0) + row * 9, *b = static_cast<const float *>(op.in1) + row * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) { rows.a[i] = a[i]; rows.b[i] = b[i]; }
    if constexpr (Op::kIn2 != 0) {
        const float *gw = static_cast<const float *>(g) + row * Op::kIn2N;
#pragma unroll
        for (int i = 0; i < Op::kIn2N; ++i) rows.c[i] = gw[i];
    }
    op.template compute<float, 1>(rows, ctx);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        d1[row * 9 + i] = rows.o0[i];
        if constexpr (Op::kOut1 != 0) d2[row * 9 + i] = rows.o1[i];
    }
}

// The same gradient from float64 data (so3_angle_bwd_f64): one row per thread, grid-stride.  g: per-row, or one shared value (scalar).
__global__ __launch_bounds__(kBlock) void k_angle_bwd_f64(const double *__restrict__ R1, const double *__restrict__ R2, const double *__restrict__ g,
                                                          bool scalar, double div, double unit, double lo, double hi, double *__restrict__ d1,
                                                          double *__restrict__ d2, int64_t B) {
#pragma clang fp contract(off)
    for (int64_t row = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; row < B; row += static_cast<int64_t>(gridDim.x) * kBlock) {
        double a[9], b[9], tr = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) { a[i] = R1[row * 9 + i]; b[i] = R2[row * 9 + i]; tr = fma(a[i], b[i], tr); }
        const double c = (tr - 1.0) * 0.5;
        const double up = (scalar ? g[0] : g[row]) / div;
        const double om = 1.0 - c * c;
        double h = (c >= lo && c <= hi && om > 0.0) ? (up * unit) * (-1.0 / __builtin_sqrt(om)) * 0.5 : 0.0;
        if (c != c) h = c;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            if (d1 != nullptr) d1[row * 9 + i] = h * b[i];
            if (d2 != nullptr) d2[row * 9 + i] = h * a[i];
        }
    }
}

// ---- K5 -------------------------------------------------------------------------------------------
// One wave per cloud at a time; lane i takes points i, i+64, ... as 12-byte (dwordx3) loads, so each
// wave-instruction reads 768 contiguous bytes.  The nine partial sums are reduced across the wave
// (DPP within rows of 16, then two cross-row exchanges); lane j of the wave keeps the H of the j-th
// cloud the wave has processed, and after its last cloud the wave runs the K1 body once with one
// cloud per lane.
__device__ __forceinline__ float dpp_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float dpp_xor2(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
}
__device__ __forceinline__ float dpp_half_mirror(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
}
__device__ __forceinline__ float dpp_mirror(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));  // row_mirror
}
__device__ __forceinline__ float wave_allsum(float v) {
    v += dpp_xor1(v);
    v += dpp_xor2(v);
    v += dpp_half_mirror(v);
    v += dpp_mirror(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
constexpr int kKabschUnroll = 8;           // 8 x 2 x 768 B = 12 KB of loads in flight per wave

__global__ __launch_bounds__(kBlock) void k_kabsch(const float *__restrict__ P, const float *__restrict__ Q,
                                                   float *__restrict__ R, float *__restrict__ H, int64_t B, int32_t N,
                                                   int clouds_per_wave) {
    const int lane = threadIdx.x & 63;
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);         // SGPR: cloud indices stay scalar
    const int64_t wave = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + wave_in_block;
    const int64_t c0 = wave * clouds_per_wave;
    if (c0 >= B) return;
    const int nc = static_cast<int>(min<int64_t>(clouds_per_wave, B - c0));
    float h[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) h[i] = (i & 3) == 0 ? 1.f : 0.f;
    const unsigned cloud_bytes = static_cast<unsigned>(N) * 12u;
    for (int j = 0; j < nc; ++j) {
        // One buffer descriptor per cloud (num_records = N*12 B): lanes past the last point read zeros, which add
        // nothing to the sums, so the point loop needs no tail and no exec-masked branch.  Streamed once: nt.
        const so3::rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P) + (c0 + j) * N * 3, 0, cloud_bytes, so3::kRsrcFlags);
        const so3::rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Q) + (c0 + j) * N * 3, 0, cloud_bytes, so3::kRsrcFlags);
        float acc[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) acc[i] = 0.f;
        for (int i0 = 0; i0 < N; i0 += 64 * kKabschUnroll) {
            u32x3 pp[kKabschUnroll], qq[kKabschUnroll];
#pragma unroll
            for (int u = 0; u < kKabschUnroll; ++u) {
                const int off = (i0 + 64 * u + lane) * 12;
                pp[u] = __builtin_amdgcn_raw_buffer_load_b96(rp, off, 0, so3::kStreamNt);
                qq[u] = __builtin_amdgcn_raw_buffer_load_b96(rq, off, 0, so3::kStreamNt);
            }
#pragma unroll
            for (int u = 0; u < kKabschUnroll; ++u) {
                const float px = __uint_as_float(pp[u].x), py = __uint_as_float(pp[u].y), pz = __uint_as_float(pp[u].z);
                const float qx = __uint_as_float(qq[u].x), qy = __uint_as_float(qq[u].y), qz = __uint_as_float(qq[u].z);
                acc[0] = fmaf(qx, px, acc[0]); acc[1] = fmaf(qx, py, acc[1]); acc[2] = fmaf(qx, pz, acc[2]);
                acc[3] = fmaf(qy, px, acc[3]); acc[4] = fmaf(qy, py, acc[4]); acc[5] = fmaf(qy, pz, acc[5]);
                acc[6] = fmaf(qz, px, acc[6]); acc[7] = fmaf(qz, py, acc[7]); acc[8] = fmaf(qz, pz, acc[8]);
            }
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const float tot = wave_allsum(acc[i]);
            h[i] = (lane == j) ? tot : h[i];
        }
    }
    const bool active = lane < nc;
    float r[9];
    so3::project_rotation<float>(h, r);
    if (active) {
        float *out = R + (c0 + lane) * 9;
#pragma unroll
        for (int i = 0; i < 9; ++i) out[i] = r[i];
        if (H != nullptr) {
            float *ho = H + (c0 + lane) * 9;
#pragma unroll
            for (int i = 0; i < 9; ++i) ho[i] = h[i];
        }
    }
}

// ---- next row f4: on-device pair synthesis for Kabsch (point_cloud/prepare.py:21-49, point_cloud/main.py:173-181) --
// (a) the reference's rotation sampler as a kernel: quaternion (cos t, axis sin t) -> matrix, given the random draws;
// (b) Kabsch with the second cloud synthesised on the fly, q_i = R_gt p_i + sigma n_i, so only P is read from HBM.
// The noise is a counter-based generator (no state): a 32-bit mix of (seed, cloud, point, pair) -> two uniforms ->
// Box-Muller; restated by the test oracle (synth_normal_np).
__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// Six standard normals for a PAIR of points from three Box-Muller pairs (round 6; rounds 2-5: two pairs per point, the second one's sine
// thrown away -- seven transcendentals and five quarter-rate integer multiplies per point, now six and three and a half):
//     point a: (r_A cos, r_A sin)(2 pi u_A), r_C cos(2 pi u_C)        point b: (r_B cos, r_B sin)(2 pi u_B), r_C sin(2 pi u_C),   r = sqrt(-2 ln u_r).
// The pair of point p is j = (p >> 7) * 64 + (p & 63) and p is its point a or b by bit 6: the two points a lane holds in neighbouring trips of
// the kernel's loop (p and p + 64), whatever the cloud's length or the loop's unrolling.  Hardware transcendentals: v_log_f32 (log2),
// v_cos_f32 / v_sin_f32 (argument in turns).
// The integer side is what the generator costs (rounds 2-4: six full mixes per point, 80 us of the kernel's 223 per 65 536 x 1024
// points; the transcendentals were the smaller half), so:
//   * ONE full mix per pair, h0 = mix(cloud's key ^ pair's counter), and four single-multiply rounds of it with different shifts,
//     multipliers and pre-whitening, h1..h4 -- pairwise 64 x 64 and 256-fold-magnified chi-square of the six uniforms, against each
//     other and against the next pair's / cloud's, stay within 3.1 sigma over 4 seeds x 4M pairs (tests/test_oracle_golden.py holds a
//     reduced form; two rounds of the SAME shape fail it at 9 sigma);
//   * a uniform is 23 bits dropped into the mantissa of a float in [1, 2) by one v_alignbit_b32: the radii take 2 - x in (0, 1], the
//     angles take x as it is (cosine and sine have period one turn);
//   * u_rA, u_A, u_rB, u_B, u_rC = the high 23 bits of h0 .. h4; u_C = the high 23 bits of the four low bytes of h0 .. h3 (bits no other
//     uniform uses).
__device__ __forceinline__ unsigned synth_cloud_key(unsigned seed, unsigned cloud) { return mix32(seed ^ mix32(cloud * 0x9e3779b9u + 0x85ebca6bu)); }
__device__ __forceinline__ float synth_unit_float(unsigned h) { return __uint_as_float(__builtin_amdgcn_alignbit(0x7fu, h, 9u)); }   // 0x3f800000 | h >> 9
__device__ __forceinline__ void synth_normal3x2(unsigned cloud_key, unsigned pair, float (&na)[3], float (&nb)[3]) {
    const unsigned h0 = mix32(cloud_key ^ (pair * 0x9e3779b9u + 0xc2b2ae35u));
    unsigned h1 = h0 + 0x27d4eb2fu; h1 ^= h1 >> 16; h1 *= 0x7feb352du; h1 ^= h1 >> 15;
    unsigned h2 = h0 ^ 0x165667b1u; h2 ^= h2 >> 15; h2 *= 0x2c1b3c6du; h2 ^= h2 >> 16;
    unsigned h3 = h0 + 0x9e3779b1u; h3 ^= h3 >> 17; h3 *= 0x297a2d39u; h3 ^= h3 >> 14;
    unsigned h4 = h0 ^ 0x85ebca77u; h4 ^= h4 >> 14; h4 *= 0xc2b2ae3du; h4 ^= h4 >> 17;
    const unsigned low = __builtin_amdgcn_perm(__builtin_amdgcn_perm(h0, h1, 0x04000c0cu), __builtin_amdgcn_perm(h2, h3, 0x0c0c0400u), 0x07060100u);
    const float ra = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(2.0f - synth_unit_float(h0)));     // -2 ln2 log2(u), u in (0, 1]
    const float rb = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(2.0f - synth_unit_float(h2)));
    const float rc = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(2.0f - synth_unit_float(h4)));
    const float ta = synth_unit_float(h1), tb = synth_unit_float(h3), tc = synth_unit_float(low);                            // [1, 2) turns
    na[0] = ra * __builtin_amdgcn_cosf(ta); na[1] = ra * __builtin_amdgcn_sinf(ta); na[2] = rc * __builtin_amdgcn_cosf(tc);
    nb[0] = rb * __builtin_amdgcn_cosf(tb); nb[1] = rb * __builtin_amdgcn_sinf(tb); nb[2] = rc * __builtin_amdgcn_sinf(tc);
}

__global__ __launch_bounds__(kBlock) void k_rotations_axis_angle(const float *__restrict__ theta, const float *__restrict__ axis,
                                                                 float *__restrict__ R, int64_t B) {
    const int64_t b = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (b >= B) return;
    const float t = theta[b];
    float ax = axis[3 * b], ay = axis[3 * b + 1], az = axis[3 * b + 2];
    const float mag = fmaxf(sqrtf(ax * ax + ay * ay + az * az), 1e-8f);        // normalize_vector, prepare.py:12-18
    ax /= mag; ay /= mag; az /= mag;
    const float sn = sinf(t), qw = cosf(t);                                        // :24,27
    const float qx = ax * sn, qy = ay * sn, qz = az * sn;                          // :28-30
    const float xx = qx * qx, yy = qy * qy, zz = qz * qz, xy = qx * qy, xz = qx * qz, yz = qy * qz;
    const float xw = qx * qw, yw = qy * qw, zw = qz * qw;
    float *o = R + 9 * b;
    o[0] = 1 - 2 * yy - 2 * zz; o[1] = 2 * xy - 2 * zw;     o[2] = 2 * xz + 2 * yw;      // :43
    o[3] = 2 * xy + 2 * zw;     o[4] = 1 - 2 * xx - 2 * zz; o[5] = 2 * yz - 2 * xw;      // :44
    o[6] = 2 * xz - 2 * yw;     o[7] = 2 * yz + 2 * xw;     o[8] = 1 - 2 * xx - 2 * yy;  // :45
}

template <bool NOISE>
__global__ __launch_bounds__(kBlock) void k_kabsch_synth(const float *__restrict__ P, const float *__restrict__ Rgt, float sigma,
                                                         unsigned seed, float *__restrict__ R, float *__restrict__ H, int64_t B,
                                                         int32_t N, int clouds_per_wave) {
    const int lane = threadIdx.x & 63;
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wave = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + wave_in_block;
    const int64_t c0 = wave * clouds_per_wave;
    if (c0 >= B) return;
    const int nc = static_cast<int>(min<int64_t>(clouds_per_wave, B - c0));
    float h[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) h[i] = (i & 3) == 0 ? 1.f : 0.f;
    const unsigned cloud_bytes = static_cast<unsigned>(N) * 12u;
    for (int j = 0; j < nc; ++j) {
        const so3::rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P) + (c0 + j) * N * 3, 0, cloud_bytes, so3::kRsrcFlags);
        float g[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) g[i] = Rgt[(c0 + j) * 9 + i];                   // wave-uniform: scalar loads
        float acc[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) acc[i] = 0.f;
        const unsigned cloud_key = synth_cloud_key(seed, static_cast<unsigned>(c0 + j));      // wave-uniform: the scalar unit's
        for (int i0 = 0; i0 < N; i0 += 64 * kKabschUnroll) {
            u32x3 pp[kKabschUnroll];
#pragma unroll
            for (int u = 0; u < kKabschUnroll; ++u)
                pp[u] = __builtin_amdgcn_raw_buffer_load_b96(rp, (i0 + 64 * u + lane) * 12, 0, so3::kStreamNt);
            static_assert(kKabschUnroll % 2 == 0, "the noise is drawn for the lane's points of two neighbouring trips at a time");
            float nz[kKabschUnroll][3];
            if constexpr (NOISE) {          // (a lane past the cloud's end draws too: its p is the range check's zero, and so is q p^T)
#pragma unroll
                for (int u = 0; u < kKabschUnroll; u += 2)              // points i0 + 64 u + lane and 64 further: pair (i0 / 128 + u / 2) * 64 + lane
                    synth_normal3x2(cloud_key, static_cast<unsigned>(((i0 >> 7) + (u >> 1)) * 64 + lane), nz[u], nz[u + 1]);
            }
#pragma unroll
            for (int u = 0; u < kKabschUnroll; ++u) {
                const float px = __uint_as_float(pp[u].x), py = __uint_as_float(pp[u].y), pz = __uint_as_float(pp[u].z);
                float qx = fmaf(g[2], pz, fmaf(g[1], py, g[0] * px));               // q = R_gt p   (main.py:176-181)
                float qy = fmaf(g[5], pz, fmaf(g[4], py, g[3] * px));
                float qz = fmaf(g[8], pz, fmaf(g[7], py, g[6] * px));
                if constexpr (NOISE) {
                    qx = fmaf(sigma, nz[u][0], qx);
                    qy = fmaf(sigma, nz[u][1], qy);
                    qz = fmaf(sigma, nz[u][2], qz);
                }
                acc[0] = fmaf(qx, px, acc[0]); acc[1] = fmaf(qx, py, acc[1]); acc[2] = fmaf(qx, pz, acc[2]);
                acc[3] = fmaf(qy, px, acc[3]); acc[4] = fmaf(qy, py, acc[4]); acc[5] = fmaf(qy, pz, acc[5]);
                acc[6] = fmaf(qz, px, acc[6]); acc[7] = fmaf(qz, py, acc[7]); acc[8] = fmaf(qz, pz, acc[8]);
            }
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const float tot = wave_allsum(acc[i]);
            h[i] = (lane == j) ? tot : h[i];
        }
    }
    const bool active = lane < nc;
    float r[9];
    so3::project_rotation<float>(h, r);
    if (active) {
        float *out = R + (c0 + lane) * 9;
#pragma unroll
        for (int i = 0; i < 9; ++i) out[i] = r[i];
        if (H != nullptr) {
            float *ho = H + (c0 + lane) * 9;
#pragma unroll
            for (int i = 0; i < 9; ++i) ho[i] = h[i];
        }
    }
}

// ---- row a7 (cloud side of the Kabsch / PointNet path) ----------------------------------------------------------
// (a) the training loop's pairing rule, point_cloud/main.py:173-181: q_i = R_b p_i for every point of cloud b, written
//     either as (B,N,3) or already transposed to the (B,3,N) layout the network consumes (`gg = pc_out.transpose(1,2)`);
// (b) pc_normalize, point_cloud/prepare.py:51-56: subtract the bounding-box centre, divide by the box diagonal.
// One wave per cloud at a time, 12-byte nt loads with a per-cloud descriptor (lanes past the end read zeros and their
// stores are dropped by the range check).
constexpr int kCloudUnroll = 8;
typedef float f32x3 __attribute__((ext_vector_type(3)));

template <bool TRANSPOSED>
__global__ __launch_bounds__(kBlock) void k_rotate_clouds(const float *__restrict__ P, const float *__restrict__ R,
                                                          float *__restrict__ out, int64_t B, int32_t N, int per_wave) {
    const int lane = threadIdx.x & 63;
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wave = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + wave_in_block;
    const int64_t c0 = wave * per_wave;
    const int nc = c0 < B ? static_cast<int>(min<int64_t>(per_wave, B - c0)) : 0;
    const unsigned cloud_bytes = static_cast<unsigned>(N) * 12u;
    for (int j = 0; j < nc; ++j) {
        const float *r = R + (c0 + j) * 9;                       // wave-uniform: scalar loads
        const float r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5], r6 = r[6], r7 = r[7], r8 = r[8];
        const so3::rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P) + (c0 + j) * N * 3, 0, cloud_bytes, so3::kRsrcFlags);
        float *ob = out + (c0 + j) * N * 3;
        const so3::rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(ob, 0, cloud_bytes, so3::kRsrcFlags);
        const so3::rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(ob, 0, static_cast<unsigned>(N) * 4u, so3::kRsrcFlags);
        const so3::rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(ob + N, 0, static_cast<unsigned>(N) * 4u, so3::kRsrcFlags);
        const so3::rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(ob + 2 * static_cast<int64_t>(N), 0, static_cast<unsigned>(N) * 4u, so3::kRsrcFlags);
        for (int i0 = 0; i0 < N; i0 += 64 * kCloudUnroll) {
            u32x3 pp[kCloudUnroll];
#pragma unroll
            for (int u = 0; u < kCloudUnroll; ++u) pp[u] = __builtin_amdgcn_raw_buffer_load_b96(rp, (i0 + 64 * u + lane) * 12, 0, so3::kStreamNt);
#pragma unroll
            for (int u = 0; u < kCloudUnroll; ++u) {
                const float px = __uint_as_float(pp[u].x), py = __uint_as_float(pp[u].y), pz = __uint_as_float(pp[u].z);
                const float qx = fmaf(r0, px, fmaf(r1, py, r2 * pz));
                const float qy = fmaf(r3, px, fmaf(r4, py, r5 * pz));
                const float qz = fmaf(r6, px, fmaf(r7, py, r8 * pz));
                const int i = i0 + 64 * u + lane;
                if (TRANSPOSED) {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(qx), rx, i * 4, 0, so3::kStreamNt);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(qy), ry, i * 4, 0, so3::kStreamNt);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(qz), rz, i * 4, 0, so3::kStreamNt);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b96(u32x3{__float_as_uint(qx), __float_as_uint(qy), __float_as_uint(qz)}, ro, i * 12, 0, so3::kStreamNt);
                }
            }
        }
    }
}

__device__ __forceinline__ float wave_allmax(float v) {
    v = fmaxf(v, dpp_xor1(v));
    v = fmaxf(v, dpp_xor2(v));
    v = fmaxf(v, dpp_half_mirror(v));
    v = fmaxf(v, dpp_mirror(v));
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

// HOLD > 0: the whole cloud (N <= 64 * HOLD points) stays in registers between the two passes: one HBM read.
template <int HOLD>
__global__ __launch_bounds__(kBlock) void k_pc_normalize(const float *__restrict__ P, float *__restrict__ out,
                                                         float *__restrict__ centroid, float *__restrict__ scale_out,
                                                         int64_t B, int32_t N, int per_wave) {
    constexpr int kU = HOLD > 0 ? HOLD : kCloudUnroll;
    const int lane = threadIdx.x & 63;
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wave = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + wave_in_block;
    const int64_t c0 = wave * per_wave;
    const int nc = c0 < B ? static_cast<int>(min<int64_t>(per_wave, B - c0)) : 0;
    const unsigned cloud_bytes = static_cast<unsigned>(N) * 12u;
    const float inf = __builtin_huge_valf();
    for (int j = 0; j < nc; ++j) {
        const so3::rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P) + (c0 + j) * N * 3, 0, cloud_bytes, so3::kRsrcFlags);
        const so3::rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + (c0 + j) * N * 3, 0, cloud_bytes, so3::kRsrcFlags);
        // pass 1: bounding box.  Out-of-range lanes read zeros, which must not enter the box: they are masked by index.
        float hi[3] = {-inf, -inf, -inf}, lo[3] = {inf, inf, inf};
        u32x3 pp[kU];
        for (int i0 = 0; i0 < N; i0 += 64 * kU) {
#pragma unroll
            for (int u = 0; u < kU; ++u) pp[u] = __builtin_amdgcn_raw_buffer_load_b96(rp, (i0 + 64 * u + lane) * 12, 0, HOLD > 0 ? so3::kStreamNt : 0);
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const bool in = i0 + 64 * u + lane < N;
                const float px = __uint_as_float(pp[u].x), py = __uint_as_float(pp[u].y), pz = __uint_as_float(pp[u].z);
                hi[0] = fmaxf(hi[0], in ? px : -inf); hi[1] = fmaxf(hi[1], in ? py : -inf); hi[2] = fmaxf(hi[2], in ? pz : -inf);
                lo[0] = fminf(lo[0], in ? px : inf);  lo[1] = fminf(lo[1], in ? py : inf);  lo[2] = fminf(lo[2], in ? pz : inf);
            }
        }
        float c[3], ext2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float mx = wave_allmax(hi[k]), mn = -wave_allmax(-lo[k]);
            c[k] = (mx + mn) * 0.5f;                                   // prepare.py:52
            const float e = (mx - c[k]) - (mn - c[k]);                 // :54 on the centred cloud
            ext2 = fmaf(e, e, ext2);
        }
        const float sc = __builtin_amdgcn_sqrtf(ext2);
        const float inv = 1.0f / sc;
        if (lane == 0) {
            if (centroid != nullptr) { centroid[(c0 + j) * 3 + 0] = c[0]; centroid[(c0 + j) * 3 + 1] = c[1]; centroid[(c0 + j) * 3 + 2] = c[2]; }
            if (scale_out != nullptr) scale_out[c0 + j] = sc;
        }
        // pass 2: from the registers when the cloud fits, otherwise a second read (mostly served by L2 / Infinity Cache)
        for (int i0 = 0; i0 < N; i0 += 64 * kU) {
            if (HOLD == 0) {
#pragma unroll
                for (int u = 0; u < kU; ++u) pp[u] = __builtin_amdgcn_raw_buffer_load_b96(rp, (i0 + 64 * u + lane) * 12, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const float qx = (__uint_as_float(pp[u].x) - c[0]) * inv, qy = (__uint_as_float(pp[u].y) - c[1]) * inv,
                            qz = (__uint_as_float(pp[u].z) - c[2]) * inv;
                __builtin_amdgcn_raw_buffer_store_b96(u32x3{__float_as_uint(qx), __float_as_uint(qy), __float_as_uint(qz)}, ro,
                                                      (i0 + 64 * u + lane) * 12, 0, so3::kStreamNt);
            }
        }
    }
}

// ---- next row f6: the ADD-L1 losses that consume calculate_T_pred's output (Iterative/loss.py:10-48) ------------
// dist_b = mean_{i,c} |(T_gt p_i - T_pred p_i)_c| = mean |dR p_i + dt|,  dR = R_gt - R_pred, dt = t_gt - t_pred, and
// its gradient  dL/dR_pred[c][j] = -k sum_i sgn(d_ic) p_ij,  dL/dt_pred[c] = -k sum_i sgn(d_ic),  k = scale / (3 N),
// in one pass over the points (the K5 skeleton: one wave per sample at a time, 12-byte loads, lane j of the wave
// keeps the results of its j-th sample).  DISENT = compute_disentangled_ADD_L1_loss: the rotation term is the same
// sum with dt = 0; the translation and depth terms do not depend on the points at all -- (|dtx| + |dty|)/3 and
// |dtz|/3 -- because the two transformed clouds differ by a constant vector.
__device__ __forceinline__ float sgn_of(float d) { return __builtin_amdgcn_fmed3f(d * 0x1p127f, -1.f, 1.f); }   // -1, 0, +1

// kAddUnroll = 12-byte loads in flight per lane: 16 for large clouds (12 KB per wave, what K5 has with two arrays),
// fewer for small ones, whose zero-filled slots would only cost arithmetic.
template <bool DISENT, int kAddUnroll>
__global__ __launch_bounds__(kBlock) void k_add_l1(const float *__restrict__ Tgt, const float *__restrict__ Tpred,
                                                   const float *__restrict__ pts, float *__restrict__ dists,
                                                   double *__restrict__ loss_sum, float *__restrict__ dT, float grad_scale,
                                                   int64_t B, int32_t N, int per_wave) {
    __shared__ double red[kBlock / 64][3];
    const int lane = threadIdx.x & 63;
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wave = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + wave_in_block;
    const int64_t c0 = wave * per_wave;
    const int nc = c0 < B ? static_cast<int>(min<int64_t>(per_wave, B - c0)) : 0;
    const float inv3n = 1.0f / (3.0f * static_cast<float>(N));
    const unsigned cloud_bytes = static_cast<unsigned>(N) * 12u;
    const int slots = ((N + 64 * kAddUnroll - 1) / (64 * kAddUnroll)) * kAddUnroll;      // loads per lane and sample
    const int valid = N > lane ? (N - lane + 63) / 64 : 0;
    const float npad = static_cast<float>(slots - valid);     // zero-filled slots of this lane: each adds |dt|, sgn(dt)
    float keep[13];                                            // lane j: dist, G (9), gt (3) of sample c0 + j
    float keep_dt[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 13; ++i) keep[i] = 0.f;
    for (int j = 0; j < nc; ++j) {
        const float *tg = Tgt + (c0 + j) * 16, *tp = Tpred + (c0 + j) * 16;      // wave-uniform: scalar loads
        float dr[9], dt[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int k = 0; k < 3; ++k) dr[3 * c + k] = tg[4 * c + k] - tp[4 * c + k];
            dt[c] = tg[4 * c + 3] - tp[4 * c + 3];
        }
        const float ax = DISENT ? 0.f : dt[0], ay = DISENT ? 0.f : dt[1], az = DISENT ? 0.f : dt[2];
        const so3::rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pts) + (c0 + j) * N * 3, 0, cloud_bytes, so3::kRsrcFlags);
        float acc[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) acc[i] = 0.f;
        for (int i0 = 0; i0 < N; i0 += 64 * kAddUnroll) {
            u32x3 pp[kAddUnroll];
#pragma unroll
            for (int u = 0; u < kAddUnroll; ++u) pp[u] = __builtin_amdgcn_raw_buffer_load_b96(rp, (i0 + 64 * u + lane) * 12, 0, so3::kStreamNt);
#pragma unroll
            for (int u = 0; u < kAddUnroll; ++u) {
                const float px = __uint_as_float(pp[u].x), py = __uint_as_float(pp[u].y), pz = __uint_as_float(pp[u].z);
                const float dx = fmaf(dr[0], px, fmaf(dr[1], py, fmaf(dr[2], pz, ax)));
                const float dy = fmaf(dr[3], px, fmaf(dr[4], py, fmaf(dr[5], pz, ay)));
                const float dz = fmaf(dr[6], px, fmaf(dr[7], py, fmaf(dr[8], pz, az)));
                acc[0] += fabsf(dx) + fabsf(dy) + fabsf(dz);
                const float sx = sgn_of(dx), sy = sgn_of(dy), sz = sgn_of(dz);
                acc[1] = fmaf(sx, px, acc[1]); acc[2] = fmaf(sx, py, acc[2]); acc[3] = fmaf(sx, pz, acc[3]);
                acc[4] = fmaf(sy, px, acc[4]); acc[5] = fmaf(sy, py, acc[5]); acc[6] = fmaf(sy, pz, acc[6]);
                acc[7] = fmaf(sz, px, acc[7]); acc[8] = fmaf(sz, py, acc[8]); acc[9] = fmaf(sz, pz, acc[9]);
                acc[10] += sx; acc[11] += sy; acc[12] += sz;
            }
        }
        if (!DISENT) {      // take the zero-filled slots back out (they saw d = dt exactly)
            acc[0] = fmaf(-npad, fabsf(ax) + fabsf(ay) + fabsf(az), acc[0]);
            acc[10] = fmaf(-npad, sgn_of(ax), acc[10]);
            acc[11] = fmaf(-npad, sgn_of(ay), acc[11]);
            acc[12] = fmaf(-npad, sgn_of(az), acc[12]);
        }
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            const float tot = wave_allsum(acc[i]);
            keep[i] = (lane == j) ? tot : keep[i];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) keep_dt[c] = (lane == j) ? dt[c] : keep_dt[c];
    }
    // lane j < nc finishes sample c0 + j
    const bool active = lane < nc;
    const float dist = keep[0] * inv3n;
    double part[3] = {0.0, 0.0, 0.0};
    if (active) {
        const int64_t b = c0 + lane;
        part[0] = dist;
        if (DISENT) {
            part[1] = (fabsf(keep_dt[0]) + fabsf(keep_dt[1])) * (1.0f / 3.0f);
            part[2] = fabsf(keep_dt[2]) * (1.0f / 3.0f);
        }
        if (dists != nullptr) dists[b] = dist;
        if (dT != nullptr) {
            const float k = -grad_scale * inv3n, kt = -grad_scale * (1.0f / 3.0f);
            float *o = dT + b * 16;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                o[4 * c + 0] = k * keep[1 + 3 * c]; o[4 * c + 1] = k * keep[2 + 3 * c]; o[4 * c + 2] = k * keep[3 + 3 * c];
                o[4 * c + 3] = DISENT ? kt * sgn_of(keep_dt[c]) : k * keep[10 + c];
            }
            o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 0.f;
        }
    }
    if (loss_sum != nullptr) {
        constexpr int kTerms = DISENT ? 3 : 1;
#pragma unroll
        for (int t = 0; t < kTerms; ++t) {
            const double v = wave_sum(part[t]);
            if (lane == 0) red[wave_in_block][t] = v;
        }
        __syncthreads();
        if (threadIdx.x < kTerms) {
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < kBlock / 64; ++w) tot += red[w][threadIdx.x];
            atomicAdd(loss_sum + threadIdx.x, tot);
        }
    }
}

// ---- float64 head and backward (the reference's functions accept double tensors): the same templates over
// T = double, one row per thread with plain loads -- a convenience path, not a benchmark configuration.
// Four fixed sweeps, then sweeps until the wave-wide residual is below 1e-14 (at most six more).
// Diagnostic twin of K1 (tests/test_gpu_parity.py: the adversarial search on the fast path's certificate): one row per thread,
// R as every forward kernel computes it (project_rotation<float>: the same IEEE operations as the packed engine) and, beside it,
// the DEVICE's own verdict -- did the fast path settle the row, or did it hand it to the Jacobi path.
__global__ __launch_bounds__(kBlock) void k_project_diag(const float *__restrict__ M, float *__restrict__ R, uint8_t *__restrict__ hard, int64_t B) {
    const int64_t row = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const bool active = row < B;
    float m[9], r[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = active ? M[row * 9 + i] : ((i & 3) == 0 ? 1.f : 0.f);
    const bool h = so3::project_rotation<float>(m, r);
    if (active) {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[row * 9 + i] = r[i];
        hard[row] = h ? 1 : 0;
    }
}

// ---- float64 arguments of the metrics and the loss (the reference's functions accept double tensors and, for the metrics,
// cast to double themselves: rotation_representation.py:232-233) -- one row per thread straight from global memory: not a
// benchmark path, but no ATen arithmetic either.  MODE 0: angle_error (float64, range flag, unit = 180/pi or 1);
// MODE 1: compute_geodesic_distance_from_two_matrices (radians, hard clamp, no flag);  MODE 2: geodesic(R1, R2, reduction)
// (point_cloud/main.py:61-73: clamp to [lo, hi], the angles' sum added onto a zeroed accumulator).
// Grid-stride (at most 2048 workgroups).  How the reduction is finished -- `how`: 0 = atomics onto accumulators an init launch has
// zeroed; 1 = ONE workgroup (B <= 1024): it writes sum, count and flag itself; 2 = caller's workspace: partial per workgroup, a ticket,
// the last one writes everything (so3_rows.h, ticket_finish).  1 and 2 are one launch per call.
template <int MODE, bool WANT_ROWS, bool WANT_SUM>
__global__ __launch_bounds__(kBlock) void k_angle_f64(const double *__restrict__ R1, const double *__restrict__ R2, double *__restrict__ out,
                                                      double *__restrict__ sum_count, int32_t *__restrict__ range_flag, double unit, int64_t B,
                                                      so3::ReduceWs *ws, int how, double lo = -1.0, double hi = 1.0) {
    __shared__ double red[4];
    double acc = 0.0;
    bool bad = false;
    for (int64_t row = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; row < B; row += static_cast<int64_t>(gridDim.x) * kBlock) {
        double tr = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) tr = fma(R1[row * 9 + i], R2[row * 9 + i], tr);
        const double c_raw = (tr - 1.0) * 0.5;
        bad |= MODE == 0 && (c_raw < -1.1 || c_raw > 1.1);
        double c = fmin(fmax(c_raw, lo), hi);
        if (c_raw != c_raw) c = c_raw;                          // clamp keeps NaN
        const double ang = so3::acos_f64(c) * unit;
        if (WANT_ROWS) out[row] = ang;
        acc += ang;
    }
    const bool any_bad = MODE == 0 && __syncthreads_or(bad ? 1 : 0) != 0;
    const double total = WANT_SUM ? block_sum(acc, red) : 0.0;
    if (MODE == 2 && WANT_SUM && threadIdx.x == 0) atomicAdd(sum_count, total);
    if (MODE != 0) return;
    if (how == 2) {
        so3::ticket_finish<kBlock>(ws, blockIdx.x, gridDim.x, gridDim.x, total, any_bad, [&](double t, bool any) {
            if (WANT_SUM) { sum_count[0] = t; sum_count[1] = static_cast<double>(B); }
            if (range_flag != nullptr) *range_flag = any ? 1 : 0;
        });
    } else if (threadIdx.x == 0) {
        if (how == 1) {
            if (WANT_SUM) { sum_count[0] = total; sum_count[1] = static_cast<double>(B); }
            if (range_flag != nullptr) *range_flag = any_bad ? 1 : 0;
        } else {
            if (WANT_SUM) atomicAdd(sum_count, total);
            if (any_bad && range_flag != nullptr) atomicOr(range_flag, 1);
        }
    }
}

// loss_frobenius in float64: sum_b ||Rtrue_b - Rpred_b||_F, optional dRpred_b = (Rpred_b - Rtrue_b) / (B ||.||_F); `how` as above
// (1 / 2: loss_sum and, if asked for, loss_mean = loss_sum / B are written by the kernel).
template <bool WANT_GRAD>
__global__ __launch_bounds__(kBlock) void k_frob_loss_f64(const double *__restrict__ Rpred, const double *__restrict__ Rtrue,
                                                          double *__restrict__ dRpred, double *__restrict__ loss_sum, double *__restrict__ loss_mean,
                                                          int64_t B, double inv_b, so3::ReduceWs *ws, int how) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int64_t row = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; row < B; row += static_cast<int64_t>(gridDim.x) * kBlock) {
        double g[9], n2 = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            g[i] = Rpred[row * 9 + i] - Rtrue[row * 9 + i];
            n2 = fma(g[i], g[i], n2);
        }
        const double nrm = __builtin_sqrt(n2);
        if (WANT_GRAD) {
            const double gs = n2 > 0.0 ? inv_b / nrm : 0.0;     // zero difference -> zero gradient (the reference: NaN)
#pragma unroll
            for (int i = 0; i < 9; ++i) dRpred[row * 9 + i] = g[i] * gs;
        }
        acc += nrm;
    }
    const double total = block_sum(acc, red);
    auto write = [&](double t) {
        *loss_sum = t;
        if (loss_mean != nullptr) *loss_mean = t * inv_b;
    };
    if (how == 2) so3::ticket_finish<kBlock>(ws, blockIdx.x, gridDim.x, gridDim.x, total, false, [&](double t, bool) { write(t); });
    else if (threadIdx.x == 0) { if (how == 1) write(total); else atomicAdd(loss_sum, total); }
}
__global__ void k_mean_from_sum_f64(const double *__restrict__ loss_sum, double *__restrict__ loss_mean, double inv_b) {
    *loss_mean = *loss_sum * inv_b;
}

template <bool BWD>
__global__ __launch_bounds__(kBlock) void k_project_f64(const double *__restrict__ M, const double *__restrict__ G,
                                                        double *__restrict__ out, uint8_t *__restrict__ flip, int64_t B) {
    const int64_t row = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const bool live = row < B;
    const int64_t rr_ = live ? row : 0;                         // idle lanes recompute row 0: keeps the sweep loop wave-uniform
    double m[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = M[rr_ * 9 + i];
    if (!BWD) {
        const auto f = so3::signed_svd<false, double, 4, true, 6>(m);
        double r[9];
        so3::rotation_from(f, r);
        if (live) {
#pragma unroll
            for (int i = 0; i < 9; ++i) out[row * 9 + i] = r[i];
            if (flip != nullptr) {
                const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
                flip[row] = det < 0.0 ? 1 : 0;
            }
        }
    } else {
        double g[9], dm[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) g[i] = G[rr_ * 9 + i];
        const auto f = so3::signed_svd<true, double, 4, true, 6>(m);
        so3::project_backward(f, g, dm);
        if (live) {
#pragma unroll
            for (int i = 0; i < 9; ++i) out[row * 9 + i] = dm[i];
        }
    }
}

// ---- 6D head, one row per thread: remainder (< 64 rows) and unaligned input of the streaming kernels ----------
template <bool BWD>
__global__ __launch_bounds__(kBlock) void k_ortho6d_rows(const float *__restrict__ X, const float *__restrict__ G,
                                                         float *__restrict__ out, int64_t B) {
    const int64_t row = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (row >= B) return;
    so3::RowCtx<1> ctx{};
    if (BWD) {
        so3::Rows<float, so3::OpOrtho6dBwd> rows;
#pragma unroll
        for (int i = 0; i < 6; ++i) rows.a[i] = X[row * 6 + i];
#pragma unroll
        for (int i = 0; i < 9; ++i) rows.b[i] = G[row * 9 + i];
        so3::OpOrtho6dBwd op;
        op.compute<float, 1>(rows, ctx);
#pragma unroll
        for (int i = 0; i < 6; ++i) out[row * 6 + i] = rows.o0[i];
    } else {
        so3::Rows<float, so3::OpOrtho6d> rows;
#pragma unroll
        for (int i = 0; i < 6; ++i) rows.a[i] = X[row * 6 + i];
        so3::OpOrtho6d op;
        op.compute<float, 1>(rows, ctx);
#pragma unroll
        for (int i = 0; i < 9; ++i) out[row * 9 + i] = rows.o0[i];
    }
}

// ---- any float32 row operation, one row per thread: remainder (< 64 rows) and unaligned input of the streaming kernels
template <class Op>
__global__ __launch_bounds__(kBlock) void k_op_rows(Op op, int64_t B) {
    const int64_t row = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (row >= B) return;
    so3::RowCtx<1> ctx{};
    so3::Rows<float, Op> rows;
    const float *a = static_cast<const float *>(op.in0) + row * Op::kIn0N;
#pragma unroll
    for (int i = 0; i < Op::kIn0N; ++i) rows.a[i] = a[i];
    if constexpr (Op::kIn1 != 0) {
        const float *b = static_cast<const float *>(op.in1) + row * Op::kIn1N;
#pragma unroll
        for (int i = 0; i < Op::kIn1N; ++i) rows.b[i] = b[i];
    }
    op.template compute<float, 1>(rows, ctx);
    float *o = static_cast<float *>(op.out0) + row * Op::kOut0N;
#pragma unroll
    for (int i = 0; i < Op::kOut0N; ++i) o[i] = rows.o0[i];
}

// ---- SE(3) update, one row per thread: remainder and unaligned input of the streaming kernels -----------------
template <bool BWD>
__global__ __launch_bounds__(kBlock) void k_se3_rows(const float *__restrict__ out12, const float *__restrict__ Tinit,
                                                     const float *__restrict__ G, float *__restrict__ res, float inv_fx,
                                                     float inv_fy, int64_t B) {
    const int64_t row = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (row >= B) return;
    so3::RowCtx<1> ctx{};
    if (BWD) {
        so3::Rows<float, so3::OpSe3UpdateBwd> rows;
#pragma unroll
        for (int i = 0; i < 12; ++i) rows.a[i] = out12[row * 12 + i];
#pragma unroll
        for (int i = 0; i < 16; ++i) { rows.b[i] = Tinit[row * 16 + i]; rows.c[i] = G[row * 16 + i]; }
        so3::OpSe3UpdateBwd op; op.inv_fx = inv_fx; op.inv_fy = inv_fy;
        op.compute<float, 1>(rows, ctx);
#pragma unroll
        for (int i = 0; i < 12; ++i) res[row * 12 + i] = rows.o0[i];
    } else {
        so3::Rows<float, so3::OpSe3Update> rows;
#pragma unroll
        for (int i = 0; i < 12; ++i) rows.a[i] = out12[row * 12 + i];
#pragma unroll
        for (int i = 0; i < 16; ++i) rows.b[i] = Tinit[row * 16 + i];
        so3::OpSe3Update op; op.inv_fx = inv_fx; op.inv_fy = inv_fy;
        op.compute<float, 1>(rows, ctx);
#pragma unroll
        for (int i = 0; i < 16; ++i) res[row * 16 + i] = rows.o0[i];
    }
}

// ---- next row f3: per-class evaluation statistics of K4's angles (3D-Pose/test_per_class.py:174-216) ----------
// Count, mean, std, max, three accuracy thresholds and an EXACT median (the two middle elements averaged, as np.median) with TWO
// passes over the rows in TWO launches (rounds 2 / 3: a radix select of eight launches, 152 us per 1M rows; round 4: four launches, 41 us):
//   1. k_stats_window (all rows): per class the sum, the sum of squares, the maximum, and a histogram of the angles over a WINDOW of
//      bins on the top bits of the float64 pattern (non-negative doubles order like their bits): sixteen bins per octave from 2^-23 to
//      2^9 degrees (sixty-four for at most four classes, Win<FINE>), one bin below and one above.  The thresholds 7.5, 15 and 30 are
//      bin edges (1.875 x 2^k), so the three accuracies and the count are sums over bins -- no atomics of their own.  Workgroup-private
//      in LDS, flushed once -- into kStatReplicas copies by the workgroup's XCD: flushing into one copy, 256 same-address atomics in a
//      row on each of a few thousand words, was the larger half of the launch (tools/stats_anatomy.py; a same-address atomic at the
//      memory side takes 30-80 ns).
//   2. k_stats_collect (all rows): first the replicas' sum and, per class, the bins holding the lower and the upper middle element
//      (every workgroup for itself); then the rows of those bins (a few per cent of a class) are compacted:
//      staged in LDS (a cursor: no global atomic, no barrier in the loop), grouped by class, and appended to the class's stretch of
//      the candidate buffer -- the histogram says exactly how long each stretch is; one atomic per workgroup and class takes a share.
//      Then the workgroup draws a ticket, and workgroup c (c < ncls; further classes wrap around) FINISHES class c once every ticket
//      is drawn: the class's candidates -- a contiguous list -- go into LDS (when more than 16 384: after the first digit has thinned
//      them out), ONE 8-bit digit is voted on, and the <= 64 candidates per selection that share it are ranked by a wave (more: further
//      digits); the class's row of the result is written.  What crosses workgroups inside the launch -- the candidates -- is written with agent-scope stores (performed past the
//      XCDs' L2s before the ticket is drawn: no release fence, which round 4 measured at 4-9 us per launch) and read with agent-scope
//      loads.  The wait is bounded: a finishing workgroup whose launch-mates do not all arrive in time (they can only be queued behind
//      another kernel: the finishers hold ncls of the device's CUs, not all) selects over the rows of its class themselves -- slow, never
//      wrong, and no wave ever waits on a condition that cannot come.
// The workspace is ZERO-FILLED ONCE by the caller and every call leaves it zeroed (the finishers clear what their class used): round
// 4's zero-fill launch in front of every call is gone.
// Exact for every input: an edge bin (zeros, denormals, angles above 512 degrees) is selected on all 64 bits, and if a workgroup's
// staging overflows (more than 4096 of its rows inside the selected bins: e.g. a million equal angles) the finishing workgroups
// select over the rows themselves (two sweeps, then from LDS).
// What does NOT pay (measured, tools/stats_anatomy.py): one launch for both passes with the rows kept in registers -- the two grid-wide
// waits it needs cost what the launch boundary does (a ticket takes 3.5 us from the last workgroup's store to the first reader's
// load); the candidates' first digit voted by the collecting workgroups into a global sub-histogram -- 74 000 same-address atomics
// on 256 words added 24 us.
constexpr int kStatFields = 8;                     // count, mean, std, max, median, acc<30, acc<15, acc<7.5
constexpr int kMaxClasses = 64;
// The window at two widths: sixteen bins per octave (the top 16 bits of the pattern) for up to kMaxClasses classes, or -- FINE, for at most
// kFineClasses classes, whose histograms then take the same LDS -- sixty-four per octave (the top 18 bits): the rows of a selected bin are
// the candidates every later step handles, and with ONE class of a million rows a sixteenth of an octave held 22 000-74 000 of them
// (the class's single finishing workgroup: 30-63 us); a sixty-fourth holds a quarter of that.
constexpr int kFineClasses = 4;
template <bool FINE> struct Win {
    static constexpr int kShift = FINE ? 46 : 48;
    static constexpr int kBase = FINE ? 0x3E80 << 2 : 0x3E80;          // (bits >> kShift) of 2^-23
    static constexpr int kBins = FINE ? 2048 : 512;
    static constexpr int kHist = kBins + 2;                            // [0]: below the window, [kHist - 1]: above it (and +inf)
    static constexpr int kRow = (kHist + 3) / 4 * 4;                   // (rows padded to whole 16-byte loads)
    // bins [0, edge) hold exactly the angles below the threshold: 30, 15 and 7.5 are 1.875 x 2^k, i.e. edges of the 1/16-octave bins
    static constexpr int kEdge30 = ((0x403E - 0x3E80) << (FINE ? 2 : 0)) + 1, kEdge15 = ((0x402E - 0x3E80) << (FINE ? 2 : 0)) + 1,
                         kEdge7p5 = ((0x401E - 0x3E80) << (FINE ? 2 : 0)) + 1;
    static __device__ __forceinline__ int bin(unsigned long long key) {
        const int top = static_cast<int>(key >> kShift);
        return top < kBase ? 0 : (top >= kBase + kBins ? kHist - 1 : top - kBase + 1);
    }
    static __device__ __forceinline__ unsigned long long top16(int b) { return static_cast<unsigned long long>(kBase + b - 1) >> (FINE ? 2 : 0); }   // of an inner bin
};
constexpr int kHistWords = kMaxClasses * Win<false>::kRow;            // one replica's histograms (either width)
static_assert(kFineClasses * Win<true>::kRow <= kHistWords, "the fine histograms fit a replica");
constexpr unsigned int kCandCap = 1u << 20;
constexpr int kStatMaxWgs = 1024;
constexpr int kStatLdsKeys = 16384;                // candidates of one class that k_stats_finish keeps in LDS (128 KB + 16 KB of tags)
// The window pass's per-class sums and histograms exist kStatReplicas times, a workgroup adds into replica (its XCD's id) % kStatReplicas:
// 256 workgroups flushing ~100 bins per class into ONE copy were 256 atomics in a row on every address -- the flush was issued 3.9 us
// into the launch and performed 9-12.6 us into it, the larger half of the pass (stamps, docs/history/profiles/r05_stats_anatomy.txt).
#ifndef SO3_STAT_REPLICAS
#define SO3_STAT_REPLICAS 4
#endif
constexpr int kStatReplicas = SO3_STAT_REPLICAS;
static_assert(kStatReplicas >= 1 && kStatReplicas <= 16 && (kStatReplicas & (kStatReplicas - 1)) == 0, "SO3_STAT_REPLICAS: a power of two (stat_replica masks the XCD id with it)");
// Layout of the caller's workspace.  `acc` and `hist` are zero-filled once by the caller and left zeroed by every call (only
// k_stats_window adds into them, only a class's own finishing workgroup clears them).  The CONTROL words -- overflow, ticket,
// class_cursor -- are put to zero by the FIRST launch of every call (k_stats_window, workgroup 0): stream order puts that after every
// workgroup of the previous call, late ones included.  Round 5 cleared them at the END of the second launch, by the last class to
// finish: a finishing workgroup whose launch-mates had not arrived within its wait then cleared the words while those mates were
// still queued, and they added to the cleared words afterwards -- the next call started with a ticket above zero and trusted a
// half-filled candidate buffer (the advisor's finding; tests/test_gpu_parity.py::test_angle_stats_survives_timed_out_waits).
struct StatWork {
    double acc[kStatReplicas][kMaxClasses][4];     // sum, sumsq, max (bits), nan_count
    unsigned int overflow, ticket;                 // bit 0: some collecting workgroup's staging overflowed, bit 2: some finishing workgroup's wait timed out; collecting workgroups that have published their candidates
    unsigned int unused, pad;
    unsigned int class_cursor[kMaxClasses];        // k_stats_collect: how much of class c's stretch of the candidate buffer is taken
    unsigned int hist[kStatReplicas][kHistWords];  // class c's bins from c * Win<FINE>::kRow on
    unsigned char tag[kCandCap];                   // bit 0: counts for the lower middle element, bit 1: for the upper
    unsigned long long cand[kCandCap];             // class c's candidates: the rows of its selected bins, in a stretch whose place and length the histogram gives
};
constexpr unsigned int kStatRegion = 4096;         // candidates one workgroup of k_stats_collect can stage in LDS (48 KB)
static_assert(kStatRegion * 256u <= kCandCap, "what 256 workgroups can stage fits the buffer");

__device__ __forceinline__ unsigned long long angle_key(double a) { return static_cast<unsigned long long>(__double_as_longlong(a < 0 ? 0.0 : a)); }

#ifdef SO3_STATS_STAMP
#define STAT_STAMP(w, idx) do { if ((blockIdx.x == 0 || blockIdx.x == 100) && threadIdx.x == 0) (w)->cand[kCandCap - 64 + (blockIdx.x ? 32 : 0) + (idx)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAT_STAMP(w, idx) do { } while (0)
#endif
constexpr int kStatLdsClasses = 16;
constexpr int kStatBlock = 1024;
// Every row once: body(angle, class) -- two rows per thread and trip: one 16-byte and one 8-byte load where both arrays allow it
// from row `head` on (mode 0 / 1: head = 0 / 1 -- views like deg[1:], cls[1:] are 8 / 4 bytes off), two scalar loads each otherwise
// (mode 2); a leading and an odd last row by themselves.
template <class F>
__device__ __forceinline__ void stats_rows(const double *__restrict__ deg, const int32_t *__restrict__ cls, int64_t B, int mode, F &&body) {
    const int64_t head = mode == 1 ? 1 : 0;
    const int64_t pairs = B > head ? (B - head) / 2 : 0;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kStatBlock;
    auto fetch = [&](int64_t i, double2 &a, int2 &c) {
        c = make_int2(0, 0);
        if (mode == 2) {
            a.x = deg[2 * i]; a.y = deg[2 * i + 1];
            if (cls) { c.x = cls[2 * i]; c.y = cls[2 * i + 1]; }
        } else {
            a = reinterpret_cast<const double2 *>(deg + head)[i];
            if (cls) c = reinterpret_cast<const int2 *>(cls + head)[i];
        }
    };
    // two trips' loads in flight per thread: a million rows are two trips of the grid, and one at a time the second trip's loads waited
    // for the first one's LDS atomics (the launch is a few microseconds long: a memory round trip is a quarter of it)
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kStatBlock + threadIdx.x; i < pairs; i += 2 * stride) {
        double2 a0, a1 = make_double2(0.0, 0.0);
        int2 c0, c1 = make_int2(-1, -1);
        const bool second = i + stride < pairs;
        fetch(i, a0, c0);
        if (second) fetch(i + stride, a1, c1);
        body(a0.x, c0.x);
        body(a0.y, c0.y);
        if (second) { body(a1.x, c1.x); body(a1.y, c1.y); }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (head == 1 && B > 0) body(deg[0], cls ? cls[0] : 0);
        if (B > head && ((B - head) & 1)) body(deg[B - 1], cls ? cls[B - 1] : 0);
    }
}

// Which replica of the sums and histograms this workgroup adds into: the id of the XCD it runs on (a speed matter only: any spread
// over the replicas gives the same totals).
__device__ __forceinline__ int stat_replica() {
    return static_cast<int>(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & (kStatReplicas - 1));        // hwreg(HW_REG_XCC_ID, 0, 4)
}

// LCLS = how many classes the workgroup's histograms hold: 16 (33 KB of counters) or all 64 the interface allows (132 KB of the CU's
// 160 KB).  The float64 sums go to 32 (8) lane-private LDS slots per class and field, the slot index fastest: a wave's 64 lanes then
// fall on 32 distinct bank pairs whatever their classes (sixteen slots were four-way conflicts: 11 us of a 30-us launch).
template <int LCLS, bool FINE>
__global__ __launch_bounds__(kStatBlock) void k_stats_window(const double *__restrict__ deg, const int32_t *__restrict__ cls, int ncls, StatWork *w,
                                                             int64_t B, int mode) {
    constexpr int kSlots = LCLS <= 16 ? 32 : 8;
    constexpr int kHistBins = Win<FINE>::kHist;
    static_assert(sizeof(unsigned int) * LCLS * kHistBins + sizeof(double) * LCLS * 4 * kSlots + 64 <= 160 * 1024, "k_stats_window: LDS budget of a gfx950 CU");
    __shared__ unsigned int sh[LCLS][kHistBins];
    __shared__ double sacc[LCLS][4][kSlots];
    STAT_STAMP(w, 16);
    if (blockIdx.x == 0) {        // the call's control words (StatWork): whatever the previous call's last workgroups left in them
        if (threadIdx.x < kMaxClasses) w->class_cursor[threadIdx.x] = 0u;
        if (threadIdx.x == kMaxClasses) { w->overflow = 0u; w->ticket = 0u; w->unused = 0u; }
    }
    for (int i = threadIdx.x; i < LCLS * kHistBins; i += kStatBlock) (&sh[0][0])[i] = 0;
    for (int i = threadIdx.x; i < ncls * 4 * kSlots; i += kStatBlock) (&sacc[0][0][0])[i] = 0.0;
    __syncthreads();
    STAT_STAMP(w, 17);
    const int slot = threadIdx.x & (kSlots - 1);
    stats_rows(deg, cls, B, mode, [&](double a, int c) {
        if (c < 0 || c >= ncls) return;
        if (a != a) { atomicAdd(&sacc[c][3][slot], 1.0); return; }
        atomicAdd(&sacc[c][0][slot], a);
        atomicAdd(&sacc[c][1][slot], a * a);
        const unsigned long long key = angle_key(a);
        atomicMax(reinterpret_cast<unsigned long long *>(&sacc[c][2][slot]), key);
        atomicAdd(&sh[c][Win<FINE>::bin(key)], 1u);
    });
    __syncthreads();
    STAT_STAMP(w, 18);
    const int rep = stat_replica();
    for (int i = threadIdx.x; i < ncls * kHistBins; i += kStatBlock) {
        const unsigned int v = sh[i / kHistBins][i % kHistBins];
        if (v) atomicAdd(&w->hist[rep][i / kHistBins * Win<FINE>::kRow + i % kHistBins], v);
    }
    for (int i = threadIdx.x; i < ncls * 4; i += kStatBlock) {
        const int c = i / 4, f = i % 4;
        if (f == 2) {
            unsigned long long m = 0;
            for (int k = 0; k < kSlots; ++k) { const unsigned long long v = static_cast<unsigned long long>(__double_as_longlong(sacc[c][2][k])); m = v > m ? v : m; }
            if (m) atomicMax(reinterpret_cast<unsigned long long *>(&w->acc[rep][c][2]), m);
        } else {
            double v = 0.0;
            for (int k = 0; k < kSlots; ++k) v += sacc[c][f][k];
            if (v != 0.0) atomicAdd(&w->acc[rep][c][f], v);
        }
    }
    STAT_STAMP(w, 19);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAT_STAMP(w, 20);
}

// What every workgroup of k_stats_collect derives for itself from the histograms (20 KB per ten classes, L2-resident): per class the
// window bins of the lower / upper middle element and the rank inside, the counts that are sums over bins, the class's stretch of
// the candidate buffer.
struct StatSel {
    int bin[2][kMaxClasses];                       // the window bin of the lower / upper middle element (-1: empty class)
    long long krem[2][kMaxClasses];                // its rank inside that bin
    double count[kMaxClasses];                     // rows of the class (NaN included)
    double acc[4][kMaxClasses];                    // sum, sum of squares, maximum, NaN count: the replicas' totals
    unsigned int below[3][kMaxClasses];            // non-NaN rows below 30 / 15 / 7.5
    unsigned int base[kMaxClasses], total[kMaxClasses];   // class c's candidates: cand[base, base + total)
};
// The replicas of the histograms are added up into LDS first (`scratch`: room for kSelChunk classes), every load a wave's 256
// consecutive bytes -- read by the lanes that scan them (nine bins apiece, 36 bytes apart, eight replicas: 72 loads of 18 cache lines
// each per class) the selection took 15 us.  Then a wave per class: both selections from one reading of the class's summed histogram,
// and how many candidates the class has in all (the rows of its one or two selected bins).
constexpr int kSelChunk = 32;
constexpr int kSelAhead = 2;
template <bool FINE>
__device__ __forceinline__ void stats_select(int ncls, const StatWork *w, StatSel &sel, unsigned int *scratch) {
    constexpr int kHistBins = Win<FINE>::kHist, kHistRow = Win<FINE>::kRow, kEdge30 = Win<FINE>::kEdge30, kEdge15 = Win<FINE>::kEdge15, kEdge7p5 = Win<FINE>::kEdge7p5;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int kPerLane = (kHistBins + 63) / 64;                         // 9 (33) bins per lane
    for (int c0 = 0; c0 < ncls; c0 += kSelChunk) {
        const int nc = ncls - c0 < kSelChunk ? ncls - c0 : kSelChunk;
        __syncthreads();
        const int quads = nc * kHistRow / 4;
        for (int i0 = threadIdx.x; i0 < quads; i0 += kSelAhead * kStatBlock) {        // 16 bytes per lane and load, kSelAhead x kStatReplicas loads in flight
            uint4 v[kSelAhead][kStatReplicas];
#pragma unroll
            for (int a = 0; a < kSelAhead; ++a) {
                const int i = i0 + a * kStatBlock;
#pragma unroll
                for (int r = 0; r < kStatReplicas; ++r) v[a][r] = reinterpret_cast<const uint4 *>(&w->hist[r][c0 * kHistRow])[i < quads ? i : i0];
            }
#pragma unroll
            for (int a = 0; a < kSelAhead; ++a) {
                const int i = i0 + a * kStatBlock;
                uint4 sum = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (int r = 0; r < kStatReplicas; ++r) { sum.x += v[a][r].x; sum.y += v[a][r].y; sum.z += v[a][r].z; sum.w += v[a][r].w; }
                if (i < quads) reinterpret_cast<uint4 *>(scratch)[i] = sum;
            }
        }
        __syncthreads();
        for (int c = c0 + wave; c < c0 + nc; c += kStatBlock / 64) {
            const unsigned int *hc = scratch + (c - c0) * kHistRow;
            // (a lane's bins are read out of LDS again where it owns a selection: held in registers, the fine width's 33 spilled)
            auto count_of = [&](int j) { const int bin = kPerLane * lane + j; return bin < kHistBins ? hc[bin] : 0u; };
            unsigned int mine = 0, b30 = 0, b15 = 0, b7 = 0;
#pragma unroll 3
            for (int j = 0; j < kPerLane; ++j) {
                const int bin = kPerLane * lane + j;
                const unsigned int hj = count_of(j);
                mine += hj;
                b30 += bin < kEdge30 ? hj : 0u; b15 += bin < kEdge15 ? hj : 0u; b7 += bin < kEdge7p5 ? hj : 0u;
            }
            unsigned int incl = mine;                                       // inclusive prefix sum over the lanes
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned int up = __shfl_up(incl, off, 64);
                if (lane >= off) incl += up;
            }
            const long long n = static_cast<long long>(__shfl(incl, 63, 64));   // the non-NaN rows of the class
            const long long before = static_cast<long long>(incl - mine);
            int bin_of[2] = {-1, -1};
            unsigned int rows_of[2] = {0u, 0u};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const long long k = n > 0 ? (t == 0 ? (n - 1) / 2 : n / 2) : 0;
                const bool owner = n > 0 && k >= before && k < before + static_cast<long long>(mine);   // exactly one lane (the counts add up to n > k)
                int d = 0;
                long long kk = k - before;
                unsigned int hd = 0;
                if (owner) {
                    for (; d < kPerLane - 1; ++d) { const unsigned int v = count_of(d); if (kk < static_cast<long long>(v)) break; kk -= v; }
                    hd = count_of(d);
                    sel.bin[t][c] = kPerLane * lane + d;
                    sel.krem[t][c] = kk;
                }
                if (n <= 0 && lane == 0) { sel.bin[t][c] = -1; sel.krem[t][c] = 0; }
                const unsigned long long who = __builtin_amdgcn_ballot_w64(owner);
                if (who != 0) {                                             // (wave-uniform)
                    const int src = __builtin_ctzll(who);
                    bin_of[t] = __shfl(kPerLane * lane + d, src, 64);
                    rows_of[t] = __shfl(hd, src, 64);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { b30 += __shfl_xor(b30, off, 64); b15 += __shfl_xor(b15, off, 64); b7 += __shfl_xor(b7, off, 64); }
            if (lane == 0) {
                double a0 = 0.0, a1 = 0.0, a3 = 0.0;
                unsigned long long a2 = 0;
                for (int r = 0; r < kStatReplicas; ++r) {
                    a0 += w->acc[r][c][0]; a1 += w->acc[r][c][1]; a3 += w->acc[r][c][3];
                    const unsigned long long v = static_cast<unsigned long long>(__double_as_longlong(w->acc[r][c][2]));
                    a2 = v > a2 ? v : a2;
                }
                sel.acc[0][c] = a0; sel.acc[1][c] = a1; sel.acc[2][c] = __longlong_as_double(static_cast<long long>(a2)); sel.acc[3][c] = a3;
                sel.count[c] = static_cast<double>(n) + a3;
                sel.below[0][c] = b30; sel.below[1][c] = b15; sel.below[2][c] = b7;
                sel.total[c] = rows_of[0] + (bin_of[1] != bin_of[0] ? rows_of[1] : 0u);
            }
        }
    }
    __syncthreads();
}

// agent-scope accesses to the candidate buffer: written and read inside ONE launch by workgroups on different XCDs
__device__ __forceinline__ void put_candidate(StatWork *w, unsigned int at, unsigned long long key, unsigned char tag) {
    __hip_atomic_store(&w->cand[at], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&w->tag[at], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long get_candidate(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned int get_tag(const unsigned char *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#ifndef SO3_STAT_SPINS
#define SO3_STAT_SPINS 16384                       // x ~0.6 us: how long a finishing workgroup waits for its launch-mates' tickets before it helps itself
#endif                                             // (0 in the test build poseestimation_amd/libso3proj_spins0.so: every wait times out at once)
constexpr unsigned int kStatSpins = SO3_STAT_SPINS;
#ifndef SO3_STAT_GRID_MULT
#define SO3_STAT_GRID_MULT 1
#endif

// One class, by the whole workgroup: the class's candidates (a contiguous stretch of the buffer; into LDS when they fit), both middle
// elements by ONE radix select of 8-bit digits -- two (prefix, rank) states side by side, they part where the two elements differ --
// then the class's row of the result (np.mean / np.std / np.max / np.median / the thresholds), and the class's part of the workspace
// back to zero.  `overflow`: the candidate buffer is not to be trusted (a staging overflow somewhere, or launch-mates that did not
// arrive): select over the rows of the class themselves.
template <bool FINE>
__device__ __forceinline__ void stats_finish_class(int c, const double *__restrict__ deg, const int32_t *__restrict__ cls, StatWork *w, int64_t B,
                                                   double *__restrict__ stats, const StatSel &sel, bool overflow_in, unsigned long long *lkey,
                                                   unsigned char *ltag, unsigned int (*hh)[256]) {
    __shared__ unsigned long long s_prefix[2];
    __shared__ long long s_k[2];
    __shared__ unsigned int s_rem[2], s_cur, s_nsurv[2];
    __shared__ unsigned long long s_surv[2][64];
    const double n = sel.count[c], nan = sel.acc[3][c], m = n - nan;
    const int bins[2] = {sel.bin[0][c], sel.bin[1][c]};
    const unsigned int total = sel.total[c];                            // the class's candidates
    const unsigned long long *cand = w->cand + sel.base[c];
    const unsigned char *ctag = w->tag + sel.base[c];
    // (a stretch that would leave the buffer can only belong to a call in which some workgroup overflowed; the flag is set then)
    const bool overflow = overflow_in || static_cast<unsigned long long>(sel.base[c]) + total > kCandCap;
    constexpr int kAhead = 8;                                           // loads in flight per thread: the workgroup is alone on its CU
    bool cached = m > 0 && !overflow && total <= static_cast<unsigned int>(kStatLdsKeys);      // (workgroup-uniform throughout)
    unsigned int held = total;                                                                // how many candidates LDS holds once cached
    __syncthreads();                                                    // (LDS of the previous phase / class is free from here)
    STAT_STAMP(w, 8);
    if (cached) {
        // (eight candidates' loads in flight per thread, as in every_far_candidate below: one at a time -- each agent-scope load followed by
        // its LDS store -- a class's 7 400 candidates took eight round trips to memory, 4.4 us)
        for (unsigned int i0 = threadIdx.x; i0 < total; i0 += kAhead * kStatBlock) {
            unsigned long long key[kAhead];
            unsigned int which[kAhead];
#pragma unroll
            for (int j = 0; j < kAhead; ++j) {
                const unsigned int i = i0 + j * kStatBlock;
                key[j] = get_candidate(cand + (i < total ? i : i0));
                which[j] = get_tag(ctag + (i < total ? i : i0));
            }
#pragma unroll
            for (int j = 0; j < kAhead; ++j) {
                const unsigned int i = i0 + j * kStatBlock;
                if (i < total) { lkey[i] = key[j]; ltag[i] = static_cast<unsigned char>(which[j]); }
            }
        }
        __syncthreads();
    }
    STAT_STAMP(w, 9);
    double middle[2] = {0.0, 0.0};
    if (m > 0) {
        const bool edge[2] = {bins[0] == 0 || bins[0] == Win<FINE>::kHist - 1, bins[1] == 0 || bins[1] == Win<FINE>::kHist - 1};
        if (threadIdx.x < 2) {
            const int t = threadIdx.x;
            s_prefix[t] = edge[t] ? 0ull : Win<FINE>::top16(bins[t]);     // (FINE: the candidates share two more bits; the first digit finds them out)
            s_k[t] = sel.krem[t][c];
        }
        __syncthreads();
        // digits from bit 56 down when an edge bin is involved (it does not fix the top 16 bits), else from bit 40; a selection whose
        // bin fixes them sits out the two top digits
        for (int shift = (edge[0] || edge[1]) ? 56 : 40; shift >= 0; shift -= 8) {
            for (int i = threadIdx.x; i < 512; i += kStatBlock) (&hh[0][0])[i] = 0;
            __syncthreads();
            const bool active[2] = {edge[0] || shift <= 40, edge[1] || shift <= 40};
            const unsigned long long prefix[2] = {s_prefix[0], s_prefix[1]};
            auto vote = [&](unsigned long long key, unsigned int which) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    if (active[t] && (which >> t & 1u) && (shift == 56 || (key >> (shift + 8)) == prefix[t])) atomicAdd(&hh[t][(key >> shift) & 0xFF], 1u);
            };
            // every candidate of the class (or, after an overflow, every row of it inside the selected bins): f(key, which selections it counts for)
            // (eight loads in flight per thread: the workgroup is alone on its CU, and one load at a time -- the votes' LDS atomics keep the
            // compiler from overlapping iterations -- made a pass over 44 000 candidates 43 round trips to memory, 20 us)
            auto every_far_candidate = [&](auto &&f) {
                if (!overflow) {
                    for (unsigned int i0 = threadIdx.x; i0 < total; i0 += kAhead * kStatBlock) {
                        unsigned long long key[kAhead];
                        unsigned int which[kAhead];
#pragma unroll
                        for (int j = 0; j < kAhead; ++j) {
                            const unsigned int i = i0 + j * kStatBlock;
                            key[j] = get_candidate(cand + (i < total ? i : i0));
                            which[j] = i < total ? get_tag(ctag + i) : 0u;
                        }
#pragma unroll
                        for (int j = 0; j < kAhead; ++j)
                            if (which[j]) f(key[j], which[j]);
                    }
                } else {
                    for (int64_t i0 = threadIdx.x; i0 < B; i0 += kAhead * kStatBlock) {
                        double a[kAhead];
                        int cc[kAhead];
#pragma unroll
                        for (int j = 0; j < kAhead; ++j) {
                            const int64_t i = i0 + j * kStatBlock;
                            a[j] = deg[i < B ? i : i0];
                            cc[j] = i < B ? (cls ? cls[i] : 0) : -1;
                        }
#pragma unroll
                        for (int j = 0; j < kAhead; ++j) {
                            if (cc[j] != c || a[j] != a[j]) continue;
                            const unsigned long long key = angle_key(a[j]);
                            const int bin = Win<FINE>::bin(key);
                            const unsigned int which = (bin == bins[0] ? 1u : 0u) | (bin == bins[1] ? 2u : 0u);
                            if (which) f(key, which);
                        }
                    }
                }
            };
            if (cached) {
                for (unsigned int i = threadIdx.x; i < held; i += kStatBlock) vote(lkey[i], ltag[i]);
            } else {
                every_far_candidate(vote);
            }
            __syncthreads();
            if (shift == 40) STAT_STAMP(w, 10);
            if (threadIdx.x < 128) {                                    // wave t: the digit holding selection t's rank -- 256 bins, four per lane, a wave prefix sum
                const int t = threadIdx.x >> 6, lane = threadIdx.x & 63;
                unsigned int h[4], mine = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) { h[j] = hh[t][4 * lane + j]; mine += h[j]; }
                unsigned int incl = mine;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned int up = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += up;
                }
                const long long before = static_cast<long long>(incl - mine), k = s_k[t];
                if (active[t] && k >= before && k < before + static_cast<long long>(mine)) {
                    long long kk = k - before;
                    unsigned int d = 0;
                    for (; d < 3; ++d) { if (kk < static_cast<long long>(h[d])) break; kk -= h[d]; }
                    s_k[t] = kk;
                    s_prefix[t] = (prefix[t] << 8) | static_cast<unsigned long long>(4 * lane + d);
                    s_rem[t] = h[d];                                    // how many candidates share the element's digits so far
                }
            }
            if (threadIdx.x == 0) s_cur = 0;
            if (threadIdx.x < 2) s_nsurv[threadIdx.x] = 0;
            __syncthreads();
            // Few enough left (a digit of 4 400 candidates leaves ~17): the candidates that share the digits chosen so far -- at most 64 per
            // selection -- are gathered, and ONE wave per selection ranks them directly (a lane per survivor, 64 comparisons each): the
            // remaining four or five digit passes, a microsecond each, are not run.
            auto rank_survivors = [&]() {
                const unsigned long long np[2] = {s_prefix[0], s_prefix[1]};
                for (unsigned int i = threadIdx.x; i < held; i += kStatBlock) {
                    const unsigned long long key = lkey[i];
                    const unsigned int which = ltag[i];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        if ((which >> t & 1u) && (key >> shift) == np[t]) s_surv[t][atomicAdd(&s_nsurv[t], 1u)] = key;     // (<= s_rem[t] <= 64 of them)
                }
                __syncthreads();
                if (threadIdx.x < 128) {
                    const int t = threadIdx.x >> 6, lane = threadIdx.x & 63;
                    const unsigned int cnt = __builtin_amdgcn_readfirstlane(s_nsurv[t]);
                    const unsigned long long mine = lane < static_cast<int>(cnt) ? s_surv[t][lane] : ~0ull;
                    const unsigned int mine_lo = static_cast<unsigned int>(mine), mine_hi = static_cast<unsigned int>(mine >> 32);
                    long long rank = 0;
                    for (unsigned int j = 0; j < cnt; ++j) {            // lane j's key through the scalar unit (out of LDS: one ~100-cycle read per step, 3 us for 30 survivors)
                        const unsigned long long other = static_cast<unsigned long long>(static_cast<unsigned int>(__builtin_amdgcn_readlane(mine_hi, j))) << 32 |
                                                         static_cast<unsigned int>(__builtin_amdgcn_readlane(mine_lo, j));
                        rank += (other < mine || (other == mine && static_cast<int>(j) < lane)) ? 1 : 0;
                    }
                    if (lane < static_cast<int>(cnt) && rank == s_k[t]) s_prefix[t] = mine;       // exactly one lane: the ranks are a permutation of 0 .. cnt-1
                }
                __syncthreads();
            };
            if (cached && shift > 0 && active[0] && active[1] && s_rem[0] <= 64u && s_rem[1] <= 64u) {
                rank_survivors();
                break;                                                  // s_prefix holds both elements, all 64 bits
            }
            // Candidates that did not fit LDS (one class of a million rows: 44 000 in its middle bin; or the rows themselves after an
            // overflow): once the digits chosen so far leave few enough, those move into LDS and the remaining digits are read there --
            // two passes over the far candidates instead of six (eight).
            if (!cached && active[0] && active[1] && shift > 0 && s_rem[0] + s_rem[1] <= static_cast<unsigned int>(kStatLdsKeys)) {
                const unsigned long long np[2] = {s_prefix[0], s_prefix[1]};
                every_far_candidate([&](unsigned long long key, unsigned int which) {
                    const unsigned int keep = ((which & 1u) && (key >> shift) == np[0] ? 1u : 0u) | ((which & 2u) && (key >> shift) == np[1] ? 2u : 0u);
                    if (keep) {
                        const unsigned int at = atomicAdd(&s_cur, 1u);
                        lkey[at] = key;                                 // (at < s_rem[0] + s_rem[1] <= kStatLdsKeys)
                        ltag[at] = static_cast<unsigned char>(keep);
                    }
                });
                __syncthreads();
                held = s_cur;
                cached = true;
                if (s_rem[0] <= 64u && s_rem[1] <= 64u) {
                    rank_survivors();
                    break;
                }
            }
        }
        STAT_STAMP(w, 11);
        middle[0] = __longlong_as_double(static_cast<long long>(s_prefix[0]));
        middle[1] = __longlong_as_double(static_cast<long long>(s_prefix[1]));
    }
    if (threadIdx.x == 0) {
        const double qnan = __longlong_as_double(0x7ff8000000000000ll);
        double *o = stats + c * kStatFields;
        const double mean = sel.acc[0][c] / m;
        o[0] = n;
        o[1] = nan > 0 ? qnan : mean;
        const double var = sel.acc[1][c] / m - mean * mean;
        o[2] = nan > 0 ? o[1] : sqrt(var > 0 ? var : 0.0);                       // np.std: population standard deviation
        o[3] = nan > 0 ? o[1] : sel.acc[2][c];
        o[4] = (nan > 0 || m <= 0) ? qnan : 0.5 * (middle[0] + middle[1]);
        o[5] = sel.below[0][c] / n; o[6] = sel.below[1][c] / n; o[7] = sel.below[2][c] / n;   // (x < t).sum() / len(x)
    }
    __syncthreads();
    // the class's part of the workspace back to zero (the next call's launches come later on the stream: plain stores)
    for (int i = threadIdx.x; i < kStatReplicas * Win<FINE>::kHist; i += kStatBlock) w->hist[i / Win<FINE>::kHist][c * Win<FINE>::kRow + i % Win<FINE>::kHist] = 0u;
    if (threadIdx.x < 4 * kStatReplicas) w->acc[threadIdx.x >> 2][c][threadIdx.x & 3] = 0.0;
}

// The rows of the selected bins: staged in LDS (a cursor: no global atomic, no barrier in the loop), then grouped by class -- a
// counting sort in LDS -- and appended to the class's stretch of the candidate buffer.  The stretches are exact: the histogram says how
// many rows each class has in its selected bins, every workgroup derives the same offsets from it, and a workgroup takes its share of
// a stretch with ONE atomic per class it holds; a finishing workgroup then reads a contiguous list.  Then the ticket, and the
// finishing of the classes this workgroup answers for (see the head of this section).
template <bool FINE>
__global__ __launch_bounds__(kStatBlock) void k_stats_collect(const double *__restrict__ deg, const int32_t *__restrict__ cls, int ncls, StatWork *w,
                                                              int64_t B, int mode, double *__restrict__ stats) {
    // the finishing phase's candidates (128 KB + 16 KB); the collecting phase stages its own rows in the front of the same arrays
    __shared__ __attribute__((aligned(16))) unsigned long long lkey[kStatLdsKeys];
    __shared__ __attribute__((aligned(8))) unsigned char ltag[kStatLdsKeys];
    __shared__ unsigned int hh[2][256];
    __shared__ StatSel sel;
    __shared__ unsigned int ccount[kMaxClasses], cstart[kMaxClasses], ccur[kMaxClasses], gbase[kMaxClasses];
    __shared__ unsigned int cur, s_state;
    static_assert(kStatRegion <= kStatLdsKeys && kStatRegion * 2 <= kStatLdsKeys, "the staging area fits the finishing phase's arrays");
    unsigned long long *skey = lkey;
    unsigned short *stag = reinterpret_cast<unsigned short *>(ltag);
    STAT_STAMP(w, 0);
    static_assert((FINE ? kFineClasses : kSelChunk) * Win<FINE>::kRow * sizeof(unsigned int) <= sizeof(unsigned long long) * kStatLdsKeys, "the selection's scratch fits the candidates' array");
    stats_select<FINE>(ncls, w, sel, reinterpret_cast<unsigned int *>(lkey));
    for (int i = threadIdx.x; i < ncls; i += kStatBlock) ccount[i] = 0;
    if (threadIdx.x == 0) cur = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int at = 0;
        for (int c = 0; c < ncls; ++c) {
            sel.base[c] = at;
            at = at + sel.total[c] < at ? 0xFFFFFFFFu : at + sel.total[c];      // (saturating: beyond the buffer nothing is written anyway)
        }
    }
    STAT_STAMP(w, 1);
    stats_rows(deg, cls, B, mode, [&](double a, int c) {
        if (c < 0 || c >= ncls || a != a) return;
        const unsigned long long key = angle_key(a);
        const int bin = Win<FINE>::bin(key);
        const unsigned int t0 = bin == sel.bin[0][c] ? 1u : 0u, t1 = bin == sel.bin[1][c] ? 1u : 0u;
        if (t0 | t1) {
            const unsigned int at = atomicAdd(&cur, 1u);
            if (at < kStatRegion) { skey[at] = key; stag[at] = static_cast<unsigned short>(c | t0 << 8 | t1 << 9); }
        }
    });
    __syncthreads();
    STAT_STAMP(w, 2);
    const unsigned int n = cur < kStatRegion ? cur : kStatRegion;
    if (threadIdx.x == 0 && cur > kStatRegion) atomicOr(&w->overflow, 1u);     // (the finishing workgroups then select over the rows themselves)
    for (unsigned int i = threadIdx.x; i < n; i += kStatBlock) atomicAdd(&ccount[stag[i] & 0xFF], 1u);
    __syncthreads();
    if (threadIdx.x < static_cast<unsigned>(ncls)) {
        const int c = threadIdx.x;
        gbase[c] = ccount[c] ? sel.base[c] + atomicAdd(&w->class_cursor[c], ccount[c]) : 0u;
    }
    if (threadIdx.x == 0) {
        unsigned int at = 0;
        for (int c = 0; c < ncls; ++c) { cstart[c] = at; ccur[c] = at; at += ccount[c]; }
    }
    __syncthreads();
    for (unsigned int i = threadIdx.x; i < n; i += kStatBlock) {
        const int c = stag[i] & 0xFF;
        const unsigned int at = gbase[c] + (atomicAdd(&ccur[c], 1u) - cstart[c]);
        if (at < kCandCap) put_candidate(w, at, skey[i], static_cast<unsigned char>(stag[i] >> 8));     // (past the buffer only after some workgroup has overflowed)
    }
    // every store of this workgroup has been performed (agent scope: past the L2) before its ticket is drawn
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    STAT_STAMP(w, 3);
    if (static_cast<int>(blockIdx.x) >= ncls) {                          // not a finishing workgroup: ticket and out
        if (threadIdx.x == 0) atomicAdd(&w->ticket, 1u);
        return;
    }
    if (threadIdx.x == 0) {
        atomicAdd(&w->ticket, 1u);
        unsigned int spins = 0;
        while (__hip_atomic_load(&w->ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && spins < kStatSpins) {
            __builtin_amdgcn_s_sleep(4);
            ++spins;
        }
        const bool all_in = __hip_atomic_load(&w->ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x;
        // A wait that timed out is made known to every finishing workgroup that comes later, BEFORE this one clears anything (the
        // returning atomic has been performed when its value is here): a launch-mate that starts after this workgroup has put its
        // class's histogram back to zero derives other stretches of the candidate buffer than everyone else did, and a finishing
        // workgroup that then finds all tickets in must not trust that buffer.
        unsigned int flags = all_in ? 0u : atomicOr(&w->overflow, 4u) | 4u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        flags |= __hip_atomic_load(&w->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_state = flags;
    }
    __syncthreads();
    STAT_STAMP(w, 4);
    const bool overflow = s_state != 0u;
    for (int c = blockIdx.x; c < ncls; c += gridDim.x) {
        stats_finish_class<FINE>(c, deg, cls, w, B, stats, sel, overflow, lkey, ltag, hh);
        __syncthreads();
        STAT_STAMP(w, 5);
    }
    // (ticket, overflow and the cursors stay as they are: the next call's first launch clears them, see StatWork)
}

inline unsigned grid_for(int64_t B) { return static_cast<unsigned>((B + kBlock - 1) / kBlock); }
inline unsigned persistent_grid(int64_t B) { const unsigned t = grid_for(B); return t < 2048u ? t : 2048u; }

// Resident waves per SIMD the two-input kernels K2 / K3 / K1+K4 are built for (K1 runs at 3): their rare Jacobi branch keeps the frames
// live across the backward (190-216 VGPRs).  Round 4 measured what three waves would buy with that branch out of the loop
// (docs/history/profiles/r04_row_number_queue_ab.txt): K2 at 168 VGPRs, spill-free, 22.10 us against 22.01 us at two -- occupancy is not what bounds it.
#ifndef SO3_BLOCK_K1
#define SO3_BLOCK_K1 256
#endif
#ifndef SO3_WPS_K1
#define SO3_WPS_K1 3
#endif
#ifndef SO3_BLOCK_K3
#define SO3_BLOCK_K3 256
#endif
#ifndef SO3_BLOCK_K14
#define SO3_BLOCK_K14 256
#endif
#ifndef SO3_WPS_K2
#define SO3_WPS_K2 2
#endif
#ifndef SO3_WPS_K3
#define SO3_WPS_K3 2
#endif
#ifndef SO3_WPS_K14
#define SO3_WPS_K14 2
#endif
#ifndef SO3_K4B_NPL                  // K4b's launch shape (the metrics' backward)
#define SO3_K4B_NPL 2
#define SO3_K4B_WPS 3
#define SO3_K4B_BLOCK 256
#endif

#define SO3_CHECK_ARGS(cond, name) \
    do { if (!(cond)) return fail(SO3_ERR_INVALID, name); } while (0)
#define SO3_MAX_B (INT64_C(2147483647) * kBlock)

// Compute units of the current device (queried once per device; 256 on an MI355X in SPX mode, fewer in CPX / DPX partitions).
inline int device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

// Launch an operation on the streaming engine: NPL matrices per lane, WPS resident waves per SIMD, BLOCK threads.
// The name of a k_rows instantiation as a profiler prints it: the runtime's own (mangled) name of the kernel behind the host
// stub, demangled, without the "void " in front and the parameter list behind.
thread_local const char *g_last_kernel = "";
template <class Op, int NPL, int WPS, int BLOCK>
const char *rows_kernel_name(hipStream_t s) {
    static const std::string name = [s] {
        const char *mangled = hipKernelNameRefByPtr(reinterpret_cast<const void *>(&so3::k_rows<Op, NPL, WPS, BLOCK, false>), s);
        if (mangled == nullptr) return std::string("so3::k_rows<?>");
        int status = 0;
        char *d = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
        std::string full = (status == 0 && d != nullptr) ? d : mangled;
        free(d);
        if (full.rfind("void ", 0) == 0) full.erase(0, 5);
        int depth = 0;                                         // cut at the '(' that opens the parameter list
        for (size_t i = 0; i < full.size(); ++i) {
            if (full[i] == '<') ++depth;
            else if (full[i] == '>') --depth;
            else if (full[i] == '(' && depth == 0) { full.erase(i); break; }
        }
        return full;
    }();
    return name.c_str();
}

template <int NPL, int WPS, int BLOCK, class Op>
void launch_rows(const Op &op, int64_t nunits, hipStream_t s) {
    constexpr int kWaves = BLOCK / 64;
    g_last_kernel = rows_kernel_name<Op, NPL, WPS, BLOCK>(s);
    const int64_t rounds = (nunits + NPL - 1) / NPL;
    const int64_t want = (rounds + kWaves - 1) / kWaves;
    int64_t cap = static_cast<int64_t>(device_cus()) * 4 * WPS / kWaves;          // CUs x 4 SIMDs x WPS wave slots
    if (Op::kFixedRounds > 0) {                                                    // not persistent: a wave per k consecutive rounds, back-filled
        const int64_t waves = (rounds + Op::kFixedRounds - 1) / Op::kFixedRounds;
        cap = (waves + kWaves - 1) / kWaves;
    }
    const dim3 grid(static_cast<unsigned>(want < cap ? want : cap)), block(BLOCK);
    hipLaunchKernelGGL((so3::k_rows<Op, NPL, WPS, BLOCK, false>), grid, block, 0, s, op, nunits, nullptr);
}

// Rows [0, 64*nunits) go to the engine when every pointer is dword aligned (every float32 array is; a bfloat16 view that
// starts at an odd row is not); the rest to the tile kernels.  The engine's 16-byte buffer loads and stores need dword
// alignment only: a view that starts at row 1 of a tensor (36 B in: 4 mod 16) streams like an aligned one -- round 1
// sent such views, e.g. the shards of an uneven split, to the one-row-per-thread kernels.
inline bool aligned4(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }
inline int64_t stream_units(int64_t B, std::initializer_list<const void *> ptrs) {
    for (const void *p : ptrs)
        if (p != nullptr && !aligned4(p)) return 0;
    if (B / so3::kUnitRows > 0x7fffffff - 4096) return 0;      // the engine numbers its rounds in 32 bits (1.3e11 rows: 5 TB of float32 input)
    return B / so3::kUnitRows;
}

// A float32 row operation with one or two inputs and one output: whole units on the streaming engine, the rest
// (and everything when a pointer is not 16-byte aligned) one row per thread.
template <int NPL, int WPS, int BLOCK, class Op>
int run_row_op(Op op, int64_t B, hipStream_t s, const char *what) {
    const int64_t nunits = stream_units(B, {op.in0, op.in1, op.out0});
    if (nunits > 0) launch_rows<NPL, WPS, BLOCK>(op, nunits, s);
    const int64_t done = nunits * so3::kUnitRows, rest = B - done;
    if (rest > 0) {
        Op t = op;
        t.in0 = static_cast<const float *>(op.in0) + done * Op::kIn0N;
        if (Op::kIn1 != 0) t.in1 = static_cast<const float *>(op.in1) + done * Op::kIn1N;
        t.out0 = static_cast<float *>(op.out0) + done * Op::kOut0N;
        hipLaunchKernelGGL((k_op_rows<Op>), dim3(grid_for(rest)), dim3(kBlock), 0, s, t, rest);
    }
    return check_launch(what);
}

template <class T> inline T *advance(T *p, int64_t elems) { return p ? p + elems : nullptr; }
inline const void *advance_bytes(const void *p, int64_t bytes) { return p ? static_cast<const char *>(p) + bytes : nullptr; }
inline void *advance_bytes(void *p, int64_t bytes) { return p ? static_cast<char *>(p) + bytes : nullptr; }

// ---- K1 --------------------------------------------------------------------------------------------
template <bool BF16>
int project_fwd(const void *M, float *R, uint8_t *flip, int64_t B, void *stream) {
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B, "so3_project_fwd: B");
    if (B == 0) return 0;
    SO3_CHECK_ARGS(M != nullptr && R != nullptr, "so3_project_fwd: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    constexpr int EB = BF16 ? 2 : 4;
    const int64_t nunits = stream_units(B, {M, R});
    if (nunits > 0) {
        // two matrices per lane, three waves per SIMD: against one matrix per lane at four / five / six / eight waves (round 3, the fast
        // path, one device): 14.6-14.9 us against 15.6 / 14.9 / 16.6 (spills) / 18.4
        if (flip) { so3::OpProject<EB, true> op; op.in0 = M; op.out0 = R; op.flip = flip; launch_rows<2, SO3_WPS_K1, SO3_BLOCK_K1>(op, nunits, s); }
        else { so3::OpProject<EB, false> op; op.in0 = M; op.out0 = R; launch_rows<2, SO3_WPS_K1, SO3_BLOCK_K1>(op, nunits, s); }
    }
    const int64_t done = nunits * so3::kUnitRows, rest = B - done;
    if (rest > 0) {
        const void *Mt = advance_bytes(M, done * 9 * EB);
        float *Rt = R + done * 9;
        uint8_t *ft = advance(flip, done);
        const bool vec = (BF16 || aligned16(Mt)) && aligned16(Rt);      // for bf16 input VEC only governs the R store
        const dim3 grid(grid_for(rest)), block(kBlock);
#define LAUNCH(VE, FL) hipLaunchKernelGGL((k_project_fwd<BF16, VE, FL>), grid, block, 0, s, Mt, Rt, ft, rest)
        if (vec) { if (flip) LAUNCH(true, true); else LAUNCH(true, false); }
        else { if (flip) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    }
    return check_launch("so3_project_fwd");
}

// ---- K2 --------------------------------------------------------------------------------------------
template <bool BF16>
int project_bwd(const void *M, const float *G, void *dM, int64_t B, void *stream) {
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B, "so3_project_bwd: B");
    if (B == 0) return 0;
    SO3_CHECK_ARGS(M != nullptr && G != nullptr && dM != nullptr, "so3_project_bwd: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    constexpr int EB = BF16 ? 2 : 4;
    const int64_t nunits = stream_units(B, {M, G, dM});
    if (nunits > 0) {
        so3::OpProjectBwd<EB> op; op.in0 = M; op.in1 = G; op.out0 = dM;
        launch_rows<2, SO3_WPS_K2, 256>(op, nunits, s);
    }
    const int64_t done = nunits * so3::kUnitRows, rest = B - done;
    if (rest > 0) {
        const void *Mt = advance_bytes(M, done * 9 * EB);
        const float *Gt = G + done * 9;
        void *dt = advance_bytes(dM, done * 9 * EB);
        const bool vec = aligned16(Gt) && (BF16 || (aligned16(Mt) && aligned16(dt)));
        const dim3 grid(grid_for(rest)), block(kBlock);
        if (vec) hipLaunchKernelGGL((k_project_bwd<BF16, true>), grid, block, 0, s, Mt, Gt, dt, rest);
        else hipLaunchKernelGGL((k_project_bwd<BF16, false>), grid, block, 0, s, Mt, Gt, dt, rest);
    }
    return check_launch("so3_project_bwd");
}

// ---- K3 --------------------------------------------------------------------------------------------
// How a reduction over a large batch is finished.  With a workspace (and whole units on the engine): the remainder kernel,
// if any, runs FIRST and fills slots [0, tile_wgs); the engine launch then takes tickets and its last workgroup writes the
// result.  Without: the accumulators are zeroed by a memset / init launch and every workgroup adds to them atomically.
inline bool use_workspace(void *workspace, int64_t nunits, unsigned tile_wgs) {
    return workspace != nullptr && nunits > 0 && tile_wgs <= 1024u;        // + at most 1024 engine workgroups <= kMaxPartials
}

template <bool BF16>
int frob(const void *M, const float *Rtrue, float *R, void *dM, double *loss_sum, float *loss_mean, void *workspace, unsigned flags, int64_t B,
         void *stream) {
    SO3_CHECK_ARGS((flags & ~static_cast<unsigned>(SO3_PREZEROED)) == 0, "so3_frob_fwd_bwd: unknown flag");
    const bool prezeroed = (flags & SO3_PREZEROED) != 0;
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B, "so3_frob_fwd_bwd: B");
    SO3_CHECK_ARGS(loss_sum != nullptr || (loss_mean != nullptr && B > 0 && B <= kSmallBatch), "so3_frob_fwd_bwd: loss_sum is null");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (B > 0 && B <= kSmallBatch) {                    // one workgroup, one launch: the kernel writes loss_sum and / or the mean itself
        SO3_CHECK_ARGS(M != nullptr && Rtrue != nullptr, "so3_frob_fwd_bwd: null pointer");
        const dim3 grid(1), block(static_cast<unsigned>((B + 63) / 64 * 64));
        const float inv = 1.0f / static_cast<float>(B);
#define SMALL(WR, WD) hipLaunchKernelGGL((k_frob_small<BF16, WR, WD>), grid, block, 0, s, M, Rtrue, R, dM, loss_sum, loss_mean, static_cast<int>(B), inv)
        if (R && dM) SMALL(true, true); else if (R) SMALL(true, false); else if (dM) SMALL(false, true); else SMALL(false, false);
#undef SMALL
        return check_launch("so3_frob_fwd_bwd");
    }
    constexpr int EB = BF16 ? 2 : 4;
    const int64_t nunits = B > 0 ? stream_units(B, {M, Rtrue, R, dM}) : 0;
    const int64_t done = nunits * so3::kUnitRows, rest = B - done;
    const unsigned tile_wgs = rest > 0 ? persistent_grid(rest) : 0u;
    so3::ReduceWs *ws = use_workspace(workspace, nunits, tile_wgs) ? static_cast<so3::ReduceWs *>(workspace) : nullptr;
    if (ws == nullptr && !prezeroed) {
        hipError_t e = hipMemsetAsync(loss_sum, 0, sizeof(double), s);
        if (e != hipSuccess) return fail((int)e, "so3_frob_fwd_bwd: memset");
    }
    if (B == 0) {
        if (loss_mean != nullptr) { hipError_t e = hipMemsetAsync(loss_mean, 0, sizeof(float), s); if (e != hipSuccess) return fail((int)e, "so3_frob_fwd_bwd: memset"); }
        return 0;
    }
    SO3_CHECK_ARGS(M != nullptr && Rtrue != nullptr, "so3_frob_fwd_bwd: null pointer");
    const float inv_b = 1.0f / static_cast<float>(B);
    if (rest > 0) {
        const void *Mt = advance_bytes(M, done * 9 * EB);
        const float *Tt = Rtrue + done * 9;
        float *Rt = advance(R, done * 9);
        void *dt = advance_bytes(dM, done * 9 * EB);
        bool vec = aligned16(Tt) && (Rt == nullptr || aligned16(Rt));
        if (!BF16) vec = vec && aligned16(Mt) && (dt == nullptr || aligned16(dt));
        const dim3 grid(tile_wgs), block(kBlock);
#define LAUNCH(VE, WR, WD) hipLaunchKernelGGL((k_frob_fwd_bwd<BF16, VE, WR, WD>), grid, block, 0, s, Mt, Tt, Rt, dt, loss_sum, rest, inv_b, ws)
#define PICK(VE) do { if (R && dM) LAUNCH(VE, true, true); else if (R) LAUNCH(VE, true, false); else if (dM) LAUNCH(VE, false, true); else LAUNCH(VE, false, false); } while (0)
        if (vec) PICK(true); else PICK(false);
#undef PICK
#undef LAUNCH
    }
    if (nunits > 0) {
#define SLAUNCH(WD, WR) do { so3::OpFrobHead<EB, WD, WR> op; op.in0 = M; op.in1 = Rtrue; op.out0 = dM; op.out1 = R; \
                             op.loss_sum = loss_sum; op.inv_b = inv_b; op.loss_mean = loss_mean; op.inv_b_f64 = 1.0 / static_cast<double>(B); \
                             op.ws = ws; op.ws_slot0 = tile_wgs; launch_rows<2, SO3_WPS_K3, SO3_BLOCK_K3>(op, nunits, s); } while (0)
        if (R && dM) SLAUNCH(true, true); else if (dM) SLAUNCH(true, false); else if (R) SLAUNCH(false, true); else SLAUNCH(false, false);
#undef SLAUNCH
    }
    if (ws == nullptr && loss_mean != nullptr) k_mean_from_sum<<<1, 1, 0, s>>>(loss_sum, loss_mean, 1.0 / static_cast<double>(B));
    return check_launch("so3_frob_fwd_bwd");
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
namespace {
template <bool DISENT>
int launch_add_l1(const float *Tgt, const float *Tpred, const float *points, float *dists, double *loss_sum, float *dTpred,
                  float grad_scale, int64_t B, int32_t N, hipStream_t s, const char *what) {
    if (loss_sum != nullptr) {
        const hipError_t e = hipMemsetAsync(loss_sum, 0, (DISENT ? 3 : 1) * sizeof(double), s);
        if (e != hipSuccess) return fail(static_cast<int>(e), what);
    }
    if (B == 0) return 0;
    int64_t per_wave = B / (static_cast<int64_t>(device_cus()) * 16);              // enough waves for 256 CUs x 16, at most 64 samples per wave
    if (per_wave < 1) per_wave = 1;
    if (per_wave > 64) per_wave = 64;
    const int64_t waves = (B + per_wave - 1) / per_wave;
    const int64_t blocks = (waves + (kBlock / 64) - 1) / (kBlock / 64);
    const dim3 grid(static_cast<unsigned>(blocks)), block(kBlock);
    const int pw = static_cast<int>(per_wave);
    if (N > 512) hipLaunchKernelGGL((k_add_l1<DISENT, 16>), grid, block, 0, s, Tgt, Tpred, points, dists, loss_sum, dTpred, grad_scale, B, N, pw);
    else if (N > 128) hipLaunchKernelGGL((k_add_l1<DISENT, 8>), grid, block, 0, s, Tgt, Tpred, points, dists, loss_sum, dTpred, grad_scale, B, N, pw);
    else hipLaunchKernelGGL((k_add_l1<DISENT, 2>), grid, block, 0, s, Tgt, Tpred, points, dists, loss_sum, dTpred, grad_scale, B, N, pw);
    return check_launch(what);
}
}  // namespace

// ---- K4b: the metrics' backward (include/so3proj.h) ------------------------------------------------------------------
namespace {
template <int GRAD, bool F64MATH, bool BOTH>
void angle_bwd_launch(const float *R1, const float *R2, const void *grad, double div, double unit, double lo, double hi, float *d1, float *d2,
                      int64_t B, hipStream_t s) {
    so3::OpAngleBwd<GRAD, F64MATH, BOTH> op;
    op.in0 = R1; op.in1 = R2; op.out0 = d1; op.out1 = d2;
    op.lo = lo; op.hi = hi; op.unit = unit; op.div = div;
    if (GRAD == 0) op.gscalar = grad; else op.in2 = grad;
    const int64_t nunits = stream_units(B, {R1, R2, GRAD != 0 ? grad : nullptr, d1, d2});
    // light arithmetic, 108-152 B per row: two rows per lane, three waves per SIMD (11 KB of LDS per wave)
    if (nunits > 0) launch_rows<SO3_K4B_NPL, SO3_K4B_WPS, SO3_K4B_BLOCK>(op, nunits, s);
    const int64_t done = nunits * so3::kUnitRows, rest = B - done;
    if (rest > 0) {
        so3::OpAngleBwd<GRAD, F64MATH, BOTH> t = op;
        t.in0 = R1 + done * 9; t.in1 = R2 + done * 9;
        const void *g = GRAD != 0 ? advance_bytes(grad, done * (F64MATH ? 8 : 4)) : nullptr;
        hipLaunchKernelGGL((k_angle_bwd_rows<so3::OpAngleBwd<GRAD, F64MATH, BOTH>>), dim3(grid_for(rest)), dim3(kBlock), 0, s, t, g, d1 + done * 9,
                           advance(d2, done * 9), rest);
    }
}
}  // namespace

extern "C" {

int so3_version(void) { return SO3PROJ_VERSION; }
const char *so3_last_error(void) { return g_err; }
const char *so3_last_kernel(void) { return g_last_kernel; }

int so3_project_fwd_f32(const float *M, float *R, uint8_t *flip, int64_t B, void *stream) {
    return project_fwd<false>(M, R, flip, B, stream);
}
int so3_project_fwd_bf16(const void *M, float *R, uint8_t *flip, int64_t B, void *stream) {
    return project_fwd<true>(M, R, flip, B, stream);
}
int so3_project_bwd_f32(const float *M, const float *G, float *dM, int64_t B, void *stream) {
    return project_bwd<false>(M, G, dM, B, stream);
}
int so3_project_bwd_bf16(const void *M, const float *G, void *dM, int64_t B, void *stream) {
    return project_bwd<true>(M, G, dM, B, stream);
}
size_t so3_reduce_workspace_bytes(void) { return sizeof(so3::ReduceWs); }
int so3_frob_fwd_bwd_v2_f32(const float *M, const float *Rtrue, float *R, float *dM, double *loss_sum, float *loss_mean, void *workspace,
                            unsigned flags, int64_t B, void *stream) {
    return frob<false>(M, Rtrue, R, dM, loss_sum, loss_mean, workspace, flags, B, stream);
}
int so3_frob_fwd_bwd_v2_bf16(const void *M, const float *Rtrue, float *R, void *dM, double *loss_sum, float *loss_mean, void *workspace,
                             unsigned flags, int64_t B, void *stream) {
    return frob<true>(M, Rtrue, R, dM, loss_sum, loss_mean, workspace, flags, B, stream);
}

int so3_frob_loss_v2_f32(const float *Rpred, const float *Rtrue, float *dRpred, double *loss_sum, float *loss_mean, void *workspace,
                         unsigned flags, int64_t B, void *stream) {
    SO3_CHECK_ARGS((flags & ~static_cast<unsigned>(SO3_PREZEROED)) == 0, "so3_frob_loss_f32: unknown flag");
    const bool prezeroed = (flags & SO3_PREZEROED) != 0;
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B, "so3_frob_loss_f32: B");
    SO3_CHECK_ARGS(loss_sum != nullptr, "so3_frob_loss_f32: loss_sum is null");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (B > 0 && B <= kSmallBatch) {                     // one workgroup, one launch: the kernel writes loss_sum (and the mean) itself
        SO3_CHECK_ARGS(Rpred != nullptr && Rtrue != nullptr, "so3_frob_loss_f32: null pointer");
        const dim3 grid(1), block(static_cast<unsigned>((B + 63) / 64 * 64));
        const float inv = 1.0f / static_cast<float>(B);
        if (dRpred) hipLaunchKernelGGL((k_frob_loss_small<true>), grid, block, 0, s, Rpred, Rtrue, dRpred, loss_sum, loss_mean, static_cast<int>(B), inv);
        else hipLaunchKernelGGL((k_frob_loss_small<false>), grid, block, 0, s, Rpred, Rtrue, dRpred, loss_sum, loss_mean, static_cast<int>(B), inv);
        return check_launch("so3_frob_loss_f32");
    }
    const int64_t nunits = B > 0 ? stream_units(B, {Rpred, Rtrue, dRpred}) : 0;
    const int64_t done = nunits * so3::kUnitRows, rest = B - done;
    const unsigned tile_wgs = rest > 0 ? persistent_grid(rest) : 0u;
    so3::ReduceWs *ws = use_workspace(workspace, nunits, tile_wgs) ? static_cast<so3::ReduceWs *>(workspace) : nullptr;
    if (ws == nullptr && !prezeroed) {
        hipError_t e = hipMemsetAsync(loss_sum, 0, sizeof(double), s);
        if (e != hipSuccess) return fail((int)e, "so3_frob_loss_f32: memset");
    }
    if (B == 0) {
        if (loss_mean != nullptr) { hipError_t e = hipMemsetAsync(loss_mean, 0, sizeof(float), s); if (e != hipSuccess) return fail((int)e, "so3_frob_loss_f32: memset"); }
        return 0;
    }
    SO3_CHECK_ARGS(Rpred != nullptr && Rtrue != nullptr, "so3_frob_loss_f32: null pointer");
    const float inv_b = 1.0f / static_cast<float>(B);
    if (rest > 0) {
        const float *Pt = Rpred + done * 9, *Tt = Rtrue + done * 9;
        float *gt = advance(dRpred, done * 9);
        const bool vec = aligned16(Pt) && aligned16(Tt) && (gt == nullptr || aligned16(gt));
        const dim3 grid(tile_wgs), block(kBlock);
#define LAUNCH(VE, WG) hipLaunchKernelGGL((k_frob_loss<VE, WG>), grid, block, 0, s, Pt, Tt, gt, loss_sum, rest, inv_b, ws)
        if (vec) { if (dRpred) LAUNCH(true, true); else LAUNCH(true, false); }
        else { if (dRpred) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    }
    if (nunits > 0) {
#define SLAUNCH(WG) do { so3::OpFrobLoss<WG> op; op.in0 = Rpred; op.in1 = Rtrue; op.out0 = dRpred; op.loss_sum = loss_sum; op.inv_b = inv_b; \
                         op.loss_mean = loss_mean; op.inv_b_f64 = 1.0 / static_cast<double>(B); op.ws = ws; op.ws_slot0 = tile_wgs; \
                         launch_rows<1, 4, 1024>(op, nunits, s); } while (0)
        if (dRpred) SLAUNCH(true); else SLAUNCH(false);
#undef SLAUNCH
    }
    if (ws == nullptr && loss_mean != nullptr) k_mean_from_sum<<<1, 1, 0, s>>>(loss_sum, loss_mean, 1.0 / static_cast<double>(B));
    return check_launch("so3_frob_loss_f32");
}
// `prezeroed`: sum_count[0] and *range_flag are zero on entry (the caller hands out fresh slots of a zero-filled pool): no
// init launch, the kernels add to them as they are and one workgroup stores the row count.
int so3_angle_error_v2(const float *R1, const float *R2, double *deg, double *sum_count, int32_t *range_flag, void *workspace, unsigned flags,
                       int64_t B, void *stream) {
    SO3_CHECK_ARGS((flags & ~static_cast<unsigned>(SO3_RADIANS | SO3_PREZEROED | SO3_EXACT_F64)) == 0, "so3_angle_error: unknown flag");
    const bool radians = (flags & SO3_RADIANS) != 0, prezeroed = (flags & SO3_PREZEROED) != 0;      // (K4 alone is float64 on every row: SO3_EXACT_F64 changes nothing)
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B, "so3_angle_error: B");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double unit = radians ? 1.0 : 57.295779513082320876798154814105;   // 180/pi
    if (B > 0 && B <= kSmallBatch && (sum_count || range_flag)) {       // one workgroup: the accumulators need no launch of their own
        SO3_CHECK_ARGS(R1 != nullptr && R2 != nullptr, "so3_angle_error: null pointer");
        const dim3 grid(1), block(static_cast<unsigned>((B + 63) / 64 * 64));
        if (deg) hipLaunchKernelGGL((k_angle_small<false, false, true>), grid, block, 0, s, R1, R2, nullptr, deg, sum_count, range_flag, unit, static_cast<int>(B));
        else hipLaunchKernelGGL((k_angle_small<false, false, false>), grid, block, 0, s, R1, R2, nullptr, deg, sum_count, range_flag, unit, static_cast<int>(B));
        return check_launch("so3_angle_error");
    }
    const int64_t nunits = B > 0 ? stream_units(B, {R1, R2, deg}) : 0;
    const int64_t done = nunits * so3::kUnitRows, rest = B - done;
    const unsigned tile_wgs = rest > 0 ? grid_for(rest) : 0u;
    so3::ReduceWs *ws = (sum_count || range_flag) && use_workspace(workspace, nunits, tile_wgs) ? static_cast<so3::ReduceWs *>(workspace) : nullptr;
    // without a workspace one tiny launch zeroes the accumulators and writes the row count (instead of two memsets + a store)
    if (ws == nullptr && (sum_count || range_flag) && !(prezeroed && B > 0)) k_angle_init<<<1, 1, 0, s>>>(sum_count, range_flag, static_cast<double>(B));
    if (B == 0) return check_launch("so3_angle_error");
    const bool store_count = prezeroed && ws == nullptr;
    if (store_count && sum_count != nullptr && nunits == 0) k_angle_init<<<1, 1, 0, s>>>(sum_count, nullptr, static_cast<double>(B));   // no engine launch to store the count
    SO3_CHECK_ARGS(R1 != nullptr && R2 != nullptr, "so3_angle_error: null pointer");
    if (rest > 0) {                                            // remainder (< 64 rows) or unaligned input
        const float *A1 = R1 + done * 9, *A2 = R2 + done * 9;
        double *dg = advance(deg, done);
        const bool vec = aligned16(A1) && aligned16(A2);
        const dim3 grid(tile_wgs), block(kBlock);
#define LAUNCH(VE, WD, WS) hipLaunchKernelGGL((k_angle_error<VE, WD, WS>), grid, block, 0, s, A1, A2, dg, sum_count, range_flag, unit, rest, ws)
#define PICK(VE) do { if (deg && sum_count) LAUNCH(VE, true, true); else if (deg) LAUNCH(VE, true, false); else if (sum_count) LAUNCH(VE, false, true); else LAUNCH(VE, false, false); } while (0)
        if (vec) PICK(true); else PICK(false);
#undef PICK
#undef LAUNCH
    }
    if (nunits > 0) {
        // 1024-thread workgroups: 256 partials (or, without a workspace, 256 same-address float64 atomics at ~9 ns) at the end
#define SLAUNCH(WD, WS) do { so3::OpAngle<WD, WS> op; op.in0 = R1; op.in1 = R2; op.deg = deg; op.sum_count = sum_count; \
                             op.range_flag = range_flag; op.unit_scale = unit; op.count = static_cast<double>(B); op.ws = ws; op.ws_slot0 = tile_wgs; \
                             op.store_count = store_count; launch_rows<1, 4, 1024>(op, nunits, s); } while (0)
        if (deg && sum_count) SLAUNCH(true, true); else if (deg) SLAUNCH(true, false); else if (sum_count) SLAUNCH(false, true); else SLAUNCH(false, false);
#undef SLAUNCH
    }
    return check_launch("so3_angle_error");
}
int so3_project_angle_error_v2_f32(const float *M, const float *Rtrue, float *R, double *deg, double *sum_count, int32_t *range_flag,
                                   void *workspace, unsigned flags, int64_t B, void *stream) {
    SO3_CHECK_ARGS((flags & ~static_cast<unsigned>(SO3_RADIANS | SO3_PREZEROED | SO3_EXACT_F64)) == 0, "so3_project_angle_error_f32: unknown flag");
    const bool radians = (flags & SO3_RADIANS) != 0, prezeroed = (flags & SO3_PREZEROED) != 0, exact = (flags & SO3_EXACT_F64) != 0;
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B, "so3_project_angle_error_f32: B");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double unit = radians ? 1.0 : 57.295779513082320876798154814105;
    if (B > 0 && B <= kSmallBatch && (sum_count || range_flag)) {       // one workgroup, one launch
        SO3_CHECK_ARGS(M != nullptr && Rtrue != nullptr, "so3_project_angle_error_f32: null pointer");
        const dim3 grid(1), block(static_cast<unsigned>((B + 63) / 64 * 64));
#define SMALL(WR, WD) hipLaunchKernelGGL((k_angle_small<true, WR, WD>), grid, block, 0, s, M, Rtrue, R, deg, sum_count, range_flag, unit, static_cast<int>(B))
        if (R && deg) SMALL(true, true); else if (R) SMALL(true, false); else if (deg) SMALL(false, true); else SMALL(false, false);
#undef SMALL
        return check_launch("so3_project_angle_error_f32");
    }
    const int64_t nunits = B > 0 ? stream_units(B, {M, Rtrue, R, deg}) : 0;
    const int64_t done = nunits * so3::kUnitRows, rest = B - done;
    const unsigned tile_wgs = rest > 0 ? grid_for(rest) : 0u;
    so3::ReduceWs *ws = (sum_count || range_flag) && use_workspace(workspace, nunits, tile_wgs) ? static_cast<so3::ReduceWs *>(workspace) : nullptr;
    if (ws == nullptr && (sum_count || range_flag) && !(prezeroed && B > 0)) k_angle_init<<<1, 1, 0, s>>>(sum_count, range_flag, static_cast<double>(B));
    if (B == 0) return check_launch("so3_project_angle_error_f32");
    const bool store_count = prezeroed && ws == nullptr;
    if (store_count && sum_count != nullptr && nunits == 0) k_angle_init<<<1, 1, 0, s>>>(sum_count, nullptr, static_cast<double>(B));
    SO3_CHECK_ARGS(M != nullptr && Rtrue != nullptr, "so3_project_angle_error_f32: null pointer");
    if (rest > 0) {
        // remainder / unaligned input: the two-kernel spelling (K1 -> R -> K4) on the tail, which needs the caller's R buffer
        SO3_CHECK_ARGS(R != nullptr, "so3_project_angle_error_f32: a < 64-row remainder or unaligned input needs the R buffer");
        const float *Mt = M + done * 9, *Tt = Rtrue + done * 9;
        float *Rt = R + done * 9;
        double *dg = advance(deg, done);
        const dim3 grid(tile_wgs), block(kBlock);
        if (aligned16(Mt) && aligned16(Rt)) hipLaunchKernelGGL((k_project_fwd<false, true, false>), grid, block, 0, s, static_cast<const void *>(Mt), Rt, nullptr, rest);
        else hipLaunchKernelGGL((k_project_fwd<false, false, false>), grid, block, 0, s, static_cast<const void *>(Mt), Rt, nullptr, rest);
#define LAUNCH(WD, WS) hipLaunchKernelGGL((k_angle_error<false, WD, WS>), grid, block, 0, s, Rt, Tt, dg, sum_count, range_flag, unit, rest, ws)
        if (deg && sum_count) LAUNCH(true, true); else if (deg) LAUNCH(true, false); else if (sum_count) LAUNCH(false, true); else LAUNCH(false, false);
#undef LAUNCH
    }
    if (nunits > 0) {
#define SLAUNCH(WR, WD, WS, F32) do { so3::OpProjectAngle<4, WR, WD, WS, F32> op; op.in0 = M; op.in1 = Rtrue; op.out0 = R; op.deg = deg; \
                                 op.sum_count = sum_count; op.range_flag = range_flag; op.unit_scale = unit; op.count = static_cast<double>(B); \
                                 op.ws = ws; op.ws_slot0 = tile_wgs; op.store_count = store_count; launch_rows<2, SO3_WPS_K14, SO3_BLOCK_K14>(op, nunits, s); } while (0)
        // the sum without per-row angles: float32 trace and acos outside the band around +-1 (so3::angle_sum_f32) unless SO3_EXACT_F64
#define PICKR(WR) do { if (deg && sum_count) SLAUNCH(WR, true, true, false); else if (deg) SLAUNCH(WR, true, false, false); \
                       else if (sum_count && !exact) SLAUNCH(WR, false, true, true); else if (sum_count) SLAUNCH(WR, false, true, false); \
                       else SLAUNCH(WR, false, false, false); } while (0)
        if (R) PICKR(true); else PICKR(false);
#undef PICKR
#undef SLAUNCH
    }
    return check_launch("so3_project_angle_error_f32");
}
int so3_project_fwd_diag_f32(const float *M, float *R, uint8_t *hard, int64_t B, void *stream) {
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B, "so3_project_fwd_diag_f32: B");
    if (B == 0) return 0;
    SO3_CHECK_ARGS(M != nullptr && R != nullptr && hard != nullptr, "so3_project_fwd_diag_f32: null pointer");
    hipLaunchKernelGGL(k_project_diag, dim3(grid_for(B)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), M, R, hard, B);
    return check_launch("so3_project_fwd_diag_f32");
}

int so3_scale_f32(const float *src, const float *factor, float *dst, int64_t n, void *stream) {
    SO3_CHECK_ARGS(n >= 0, "so3_scale_f32: n");
    if (n == 0) return 0;
    SO3_CHECK_ARGS(src != nullptr && factor != nullptr && dst != nullptr, "so3_scale_f32: null pointer");
    hipLaunchKernelGGL((k_scale<false>), dim3(static_cast<unsigned>((n + 4 * kBlock - 1) / (4 * kBlock))), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       static_cast<const void *>(src), factor, static_cast<void *>(dst), n);
    return check_launch("so3_scale_f32");
}
int so3_scale_bf16(const void *src, const float *factor, void *dst, int64_t n, void *stream) {
    SO3_CHECK_ARGS(n >= 0, "so3_scale_bf16: n");
    if (n == 0) return 0;
    SO3_CHECK_ARGS(src != nullptr && factor != nullptr && dst != nullptr, "so3_scale_bf16: null pointer");
    hipLaunchKernelGGL((k_scale<true>), dim3(static_cast<unsigned>((n + 8 * kBlock - 1) / (8 * kBlock))), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       src, factor, dst, n);
    return check_launch("so3_scale_bf16");
}

// one launch for B <= 1024 (one workgroup) and with a workspace; else init / memset launch + kernel (+ the mean's)
static int reduce_how(void *workspace, int64_t B, unsigned *grid) {
    // two workgroups' worth per CU, grid-stride: the ticket at the end is one same-address atomic per workgroup (~12 ns each,
    // serialised at the memory side -- 2048 of them were 7 us on top of a 37-us kernel)
    const unsigned want = grid_for(B), cap = 2u * static_cast<unsigned>(device_cus());
    *grid = B <= kSmallBatch ? 1u : (want < cap ? want : cap);
    return B <= kSmallBatch ? 1 : (workspace != nullptr ? 2 : 0);
}

int so3_angle_error_v2_f64(const double *R1, const double *R2, double *deg, double *sum_count, int32_t *range_flag, void *workspace,
                           unsigned flags, int64_t B, void *stream) {
    SO3_CHECK_ARGS((flags & ~static_cast<unsigned>(SO3_RADIANS)) == 0, "so3_angle_error_v2_f64: unknown flag (SO3_RADIANS only)");
    const bool radians = (flags & SO3_RADIANS) != 0;
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B, "so3_angle_error_v2_f64: B");
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned grid = 1;
    const int how = B > 0 ? reduce_how(workspace, B, &grid) : 0;
    if (how == 0 && (sum_count || range_flag)) k_angle_init<<<1, 1, 0, s>>>(sum_count, range_flag, static_cast<double>(B));
    if (B == 0) return check_launch("so3_angle_error_v2_f64");
    SO3_CHECK_ARGS(R1 != nullptr && R2 != nullptr, "so3_angle_error_v2_f64: null pointer");
    const double unit = radians ? 1.0 : 57.295779513082320876798154814105;
    const dim3 block(kBlock);
    so3::ReduceWs *ws = static_cast<so3::ReduceWs *>(workspace);
#define LAUNCH(WD, WS) hipLaunchKernelGGL((k_angle_f64<0, WD, WS>), dim3(grid), block, 0, s, R1, R2, deg, sum_count, range_flag, unit, B, ws, how)
    if (deg && sum_count) LAUNCH(true, true); else if (deg) LAUNCH(true, false); else if (sum_count) LAUNCH(false, true); else LAUNCH(false, false);
#undef LAUNCH
    return check_launch("so3_angle_error_v2_f64");
}

int so3_geodesic_f64(const double *R1, const double *R2, double *theta, int64_t B, void *stream) {
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B, "so3_geodesic_f64: B");
    if (B == 0) return 0;
    SO3_CHECK_ARGS(R1 != nullptr && R2 != nullptr && theta != nullptr, "so3_geodesic_f64: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned want = grid_for(B);
    hipLaunchKernelGGL((k_angle_f64<1, true, false>), dim3(want < 2048u ? want : 2048u), dim3(kBlock), 0, s, R1, R2, theta, nullptr, nullptr, 1.0, B, nullptr, 0);
    return check_launch("so3_geodesic_f64");
}

int so3_frob_loss_v2_f64(const double *Rpred, const double *Rtrue, double *dRpred, double *loss_sum, double *loss_mean, void *workspace,
                         unsigned flags, int64_t B, void *stream) {
    SO3_CHECK_ARGS(flags == 0, "so3_frob_loss_v2_f64: unknown flag (none is defined for it)");
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B, "so3_frob_loss_v2_f64: B");
    SO3_CHECK_ARGS(loss_sum != nullptr, "so3_frob_loss_v2_f64: loss_sum is null");
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned grid = 1;
    const int how = B > 0 ? reduce_how(workspace, B, &grid) : 0;
    if (how == 0) {
        hipError_t e = hipMemsetAsync(loss_sum, 0, sizeof(double), s);
        if (e != hipSuccess) return fail((int)e, "so3_frob_loss_v2_f64: memset");
        if (B == 0) {
            if (loss_mean != nullptr) { e = hipMemsetAsync(loss_mean, 0, sizeof(double), s); if (e != hipSuccess) return fail((int)e, "so3_frob_loss_v2_f64: memset"); }
            return 0;
        }
    }
    SO3_CHECK_ARGS(Rpred != nullptr && Rtrue != nullptr, "so3_frob_loss_v2_f64: null pointer");
    const double inv_b = 1.0 / static_cast<double>(B);
    const dim3 block(kBlock);
    so3::ReduceWs *ws = static_cast<so3::ReduceWs *>(workspace);
    if (dRpred) hipLaunchKernelGGL((k_frob_loss_f64<true>), dim3(grid), block, 0, s, Rpred, Rtrue, dRpred, loss_sum, loss_mean, B, inv_b, ws, how);
    else hipLaunchKernelGGL((k_frob_loss_f64<false>), dim3(grid), block, 0, s, Rpred, Rtrue, dRpred, loss_sum, loss_mean, B, inv_b, ws, how);
    if (how == 0 && loss_mean != nullptr) k_mean_from_sum_f64<<<1, 1, 0, s>>>(loss_sum, loss_mean, inv_b);
    return check_launch("so3_frob_loss_v2_f64");
}

static int geodesic_f32(const float *R1, const float *R2, float *theta, double *sum, float *result, int mean, float eps, void *workspace, int64_t B,
                        void *stream, const char *what) {
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B && eps >= 0.f && eps < 1.f, what);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t nunits = B > 0 ? stream_units(B, {R1, R2, theta}) : 0;
    const int64_t done = nunits * so3::kUnitRows, rest = B - done;
    const unsigned tile_wgs = rest > 0 ? grid_for(rest) : 0u;
    // with a workspace: ONE launch (two with a remainder) -- the last workgroup writes the sum and the reduced result
    so3::ReduceWs *ws = sum != nullptr && use_workspace(workspace, nunits, tile_wgs) ? static_cast<so3::ReduceWs *>(workspace) : nullptr;
    if (sum != nullptr && ws == nullptr) {
        const hipError_t e = hipMemsetAsync(sum, 0, sizeof(double), s);
        if (e != hipSuccess) return fail(static_cast<int>(e), what);
    }
    const double scale = mean ? 1.0 / static_cast<double>(B) : 1.0;
    if (B > 0) {
        SO3_CHECK_ARGS(R1 != nullptr && R2 != nullptr && (theta != nullptr || sum != nullptr), what);
        const float lo = -1.f + eps, hi = 1.f - eps;          // float32 arithmetic, like torch.clamp's scalars on a float32 tensor
        if (rest > 0) {                                       // (first: its partials are in the workspace when the engine's last workgroup sums)
            const float *A1 = R1 + done * 9, *A2 = R2 + done * 9;
            const dim3 grid(tile_wgs), block(kBlock);
            if (aligned16(A1) && aligned16(A2)) hipLaunchKernelGGL((k_geodesic_f32<true>), grid, block, 0, s, A1, A2, advance(theta, done), sum, lo, hi, rest, ws);
            else hipLaunchKernelGGL((k_geodesic_f32<false>), grid, block, 0, s, A1, A2, advance(theta, done), sum, lo, hi, rest, ws);
        }
        if (nunits > 0) {
            if (sum) {
                so3::OpGeodesic<true> op; op.in0 = R1; op.in1 = R2; op.theta = theta; op.sum = sum; op.lo = lo; op.hi = hi;
                op.ws = ws; op.ws_slot0 = tile_wgs; op.result = result; op.scale = scale;
                // 1024-thread workgroups as K4: 256 partials / same-address atomics at the end, not 768 (at ~12 ns each the
                // <2, 3, 256> shape spent 5 us of its 18.5 queueing on one address: docs/history/profiles/r04_all_kernels_stats.csv)
                launch_rows<1, 4, 1024>(op, nunits, s);
            } else { so3::OpGeodesic<false> op; op.in0 = R1; op.in1 = R2; op.theta = theta; op.lo = lo; op.hi = hi; launch_rows<2, 3, 256>(op, nunits, s); }
        }
    }
    // (torch's mean of an empty tensor is NaN, its sum 0: 0 * inf / 0 * 1)
    if (result != nullptr && ws == nullptr) k_mean_from_sum<<<1, 1, 0, s>>>(sum, result, scale);
    return check_launch(what);
}
int so3_geodesic_f32(const float *R1, const float *R2, float *theta, int64_t B, void *stream) {
    if (B == 0) return 0;
    SO3_CHECK_ARGS(theta != nullptr, "so3_geodesic_f32: null pointer");
    return geodesic_f32(R1, R2, theta, nullptr, nullptr, 0, 0.f, nullptr, B, stream, "so3_geodesic_f32");
}
int so3_geodesic_eps_f32(const float *R1, const float *R2, float *theta, double *sum, float *result, int mean, float eps, void *workspace, int64_t B,
                         void *stream) {
    SO3_CHECK_ARGS(result == nullptr || sum != nullptr, "so3_geodesic_eps_f32: result needs the float64 scratch `sum`");
    return geodesic_f32(R1, R2, theta, sum, result, mean, eps, workspace, B, stream, "so3_geodesic_eps_f32");
}

int so3_angle_bwd_f32(const float *R1, const float *R2, const void *grad, double grad_div, double eps, unsigned flags, float *dR1, float *dR2,
                      int64_t B, void *stream) {
    SO3_CHECK_ARGS((flags & ~static_cast<unsigned>(SO3_RADIANS | SO3_GRAD_SCALAR | SO3_F64_MATH)) == 0, "so3_angle_bwd_f32: unknown flag");
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B, "so3_angle_bwd_f32: B");
    SO3_CHECK_ARGS(eps >= 0.0 && eps < 1.0 && grad_div != 0.0, "so3_angle_bwd_f32: eps / grad_div");
    if (B == 0) return 0;
    SO3_CHECK_ARGS(R1 != nullptr && R2 != nullptr && grad != nullptr && (dR1 != nullptr || dR2 != nullptr), "so3_angle_bwd_f32: null pointer");
    const bool f64 = (flags & SO3_F64_MATH) != 0, scalar = (flags & SO3_GRAD_SCALAR) != 0;
    SO3_CHECK_ARGS((reinterpret_cast<uintptr_t>(grad) & (f64 ? 7u : 3u)) == 0, "so3_angle_bwd_f32: grad is not aligned to its element");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double unit = (flags & SO3_RADIANS) != 0 ? 1.0 : 57.295779513082320876798154814105;
    // the clamp's bounds in the arithmetic of the spelling (torch.clamp's scalars on a float32 tensor are float32)
    const double lo = f64 ? -1.0 + eps : static_cast<double>(-1.f + static_cast<float>(eps));
    const double hi = f64 ? 1.0 - eps : static_cast<double>(1.f - static_cast<float>(eps));
    // one gradient alone: tr(R1 R2^T) is symmetric in its arguments, so dR2 is dR1 of the swapped pair
    const bool both = dR1 != nullptr && dR2 != nullptr;
    const float *A = dR1 != nullptr ? R1 : R2, *Bm = dR1 != nullptr ? R2 : R1;
    float *d1 = dR1 != nullptr ? dR1 : dR2, *d2 = both ? dR2 : nullptr;
#define GO(GR, F6, BO) angle_bwd_launch<GR, F6, BO>(A, Bm, grad, grad_div, unit, lo, hi, d1, d2, B, s)
#define PICK(F6, PER) do { if (scalar) { if (both) GO(0, F6, true); else GO(0, F6, false); } \
                           else { if (both) GO(PER, F6, true); else GO(PER, F6, false); } } while (0)
    if (f64) PICK(true, 2); else PICK(false, 1);
#undef PICK
#undef GO
    return check_launch("so3_angle_bwd_f32");
}

int so3_angle_bwd_f64(const double *R1, const double *R2, const double *grad, double grad_div, double eps, unsigned flags, double *dR1, double *dR2,
                      int64_t B, void *stream) {
    SO3_CHECK_ARGS((flags & ~static_cast<unsigned>(SO3_RADIANS | SO3_GRAD_SCALAR)) == 0, "so3_angle_bwd_f64: unknown flag");
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B, "so3_angle_bwd_f64: B");
    SO3_CHECK_ARGS(eps >= 0.0 && eps < 1.0 && grad_div != 0.0, "so3_angle_bwd_f64: eps / grad_div");
    if (B == 0) return 0;
    SO3_CHECK_ARGS(R1 != nullptr && R2 != nullptr && grad != nullptr && (dR1 != nullptr || dR2 != nullptr), "so3_angle_bwd_f64: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double unit = (flags & SO3_RADIANS) != 0 ? 1.0 : 57.295779513082320876798154814105;
    const unsigned want = grid_for(B);
    hipLaunchKernelGGL(k_angle_bwd_f64, dim3(want < 2048u ? want : 2048u), dim3(kBlock), 0, s, R1, R2, grad, (flags & SO3_GRAD_SCALAR) != 0, grad_div,
                       unit, -1.0 + eps, 1.0 - eps, dR1, dR2, B);
    return check_launch("so3_angle_bwd_f64");
}

int so3_geodesic_eps_f64(const double *R1, const double *R2, double *theta, double *sum, double *result, int mean, double eps, int64_t B,
                         void *stream) {
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B && eps >= 0.0 && eps < 1.0, "so3_geodesic_eps_f64: B / eps");
    SO3_CHECK_ARGS(result == nullptr || sum != nullptr, "so3_geodesic_eps_f64: result needs the scratch `sum`");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (sum != nullptr) {
        const hipError_t e = hipMemsetAsync(sum, 0, sizeof(double), s);
        if (e != hipSuccess) return fail(static_cast<int>(e), "so3_geodesic_eps_f64: memset");
    }
    if (B > 0) {
        SO3_CHECK_ARGS(R1 != nullptr && R2 != nullptr && (theta != nullptr || sum != nullptr), "so3_geodesic_eps_f64: null pointer");
        const unsigned want = grid_for(B);
        const dim3 grid(want < 2048u ? want : 2048u), block(kBlock);
        if (theta && sum) hipLaunchKernelGGL((k_angle_f64<2, true, true>), grid, block, 0, s, R1, R2, theta, sum, nullptr, 1.0, B, nullptr, 0, -1.0 + eps, 1.0 - eps);
        else if (theta) hipLaunchKernelGGL((k_angle_f64<2, true, false>), grid, block, 0, s, R1, R2, theta, sum, nullptr, 1.0, B, nullptr, 0, -1.0 + eps, 1.0 - eps);
        else hipLaunchKernelGGL((k_angle_f64<2, false, true>), grid, block, 0, s, R1, R2, theta, sum, nullptr, 1.0, B, nullptr, 0, -1.0 + eps, 1.0 - eps);
    }
    // (torch's mean of an empty tensor is NaN, its sum 0)
    if (result != nullptr) k_mean_from_sum_f64<<<1, 1, 0, s>>>(sum, result, mean ? 1.0 / static_cast<double>(B) : 1.0);
    return check_launch("so3_geodesic_eps_f64");
}

int so3_ortho6d_fwd_f32(const float *X, float *R, int64_t B, void *stream) {
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B, "so3_ortho6d_fwd_f32: B");
    if (B == 0) return 0;
    SO3_CHECK_ARGS(X != nullptr && R != nullptr, "so3_ortho6d_fwd_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t nunits = stream_units(B, {X, R});
    if (nunits > 0) { so3::OpOrtho6d op; op.in0 = X; op.out0 = R; launch_rows<2, 4, 256>(op, nunits, s); }
    const int64_t done = nunits * so3::kUnitRows, rest = B - done;
    if (rest > 0) hipLaunchKernelGGL((k_ortho6d_rows<false>), dim3(grid_for(rest)), dim3(kBlock), 0, s, X + done * 6, nullptr, R + done * 9, rest);
    return check_launch("so3_ortho6d_fwd_f32");
}

int so3_ortho6d_bwd_f32(const float *X, const float *G, float *dX, int64_t B, void *stream) {
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B, "so3_ortho6d_bwd_f32: B");
    if (B == 0) return 0;
    SO3_CHECK_ARGS(X != nullptr && G != nullptr && dX != nullptr, "so3_ortho6d_bwd_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t nunits = stream_units(B, {X, G, dX});
    if (nunits > 0) { so3::OpOrtho6dBwd op; op.in0 = X; op.in1 = G; op.out0 = dX; launch_rows<2, 4, 256>(op, nunits, s); }
    const int64_t done = nunits * so3::kUnitRows, rest = B - done;
    if (rest > 0) hipLaunchKernelGGL((k_ortho6d_rows<true>), dim3(grid_for(rest)), dim3(kBlock), 0, s, X + done * 6, G + done * 9, dX + done * 6, rest);
    return check_launch("so3_ortho6d_bwd_f32");
}

// The other heads of the reference's dispatch tables (include/so3proj.h, "next row f5").
#define SO3_DEFINE_HEAD(NAME, OP)                                                                                   \
    int so3_##NAME##_fwd_f32(const float *X, float *R, int64_t B, void *stream) {                                    \
        SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B, "so3_" #NAME "_fwd_f32: B");                                        \
        if (B == 0) return 0;                                                                                        \
        SO3_CHECK_ARGS(X != nullptr && R != nullptr, "so3_" #NAME "_fwd_f32: null pointer");                         \
        so3::OP<false> op; op.in0 = X; op.out0 = R;                                                                  \
        return run_row_op<1, 6, 256>(op, B, static_cast<hipStream_t>(stream), "so3_" #NAME "_fwd_f32");              \
    }                                                                                                                \
    int so3_##NAME##_bwd_f32(const float *X, const float *G, float *dX, int64_t B, void *stream) {                   \
        SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B, "so3_" #NAME "_bwd_f32: B");                                        \
        if (B == 0) return 0;                                                                                        \
        SO3_CHECK_ARGS(X != nullptr && G != nullptr && dX != nullptr, "so3_" #NAME "_bwd_f32: null pointer");        \
        so3::OP<true> op; op.in0 = X; op.in1 = G; op.out0 = dX;                                                      \
        return run_row_op<1, 6, 256>(op, B, static_cast<hipStream_t>(stream), "so3_" #NAME "_bwd_f32");              \
    }
SO3_DEFINE_HEAD(quat, OpQuat)
SO3_DEFINE_HEAD(euler, OpEuler)
SO3_DEFINE_HEAD(ortho5d, OpOrtho5d)
SO3_DEFINE_HEAD(expmap, OpExpMap)
#undef SO3_DEFINE_HEAD

int so3_add_l1_f32(const float *Tgt, const float *Tpred, const float *points, float *dists, double *loss_sum, float *dTpred,
                   float grad_scale, int64_t B, int32_t N, void *stream) {
    SO3_CHECK_ARGS(B >= 0 && B <= (INT64_C(1) << 31) && N >= 1 && N <= 150000000, "so3_add_l1_f32: B/N");
    SO3_CHECK_ARGS(B == 0 || (Tgt != nullptr && Tpred != nullptr && points != nullptr), "so3_add_l1_f32: null pointer");
    return launch_add_l1<false>(Tgt, Tpred, points, dists, loss_sum, dTpred, grad_scale, B, N, static_cast<hipStream_t>(stream), "so3_add_l1_f32");
}

int so3_add_l1_disentangled_f32(const float *Tpred, const float *Tgt, const float *points, double *loss_sum, float *dTpred,
                                float grad_scale, int64_t B, int32_t N, void *stream) {
    SO3_CHECK_ARGS(B >= 0 && B <= (INT64_C(1) << 31) && N >= 1 && N <= 150000000, "so3_add_l1_disentangled_f32: B/N");
    SO3_CHECK_ARGS(B == 0 || (Tgt != nullptr && Tpred != nullptr && points != nullptr), "so3_add_l1_disentangled_f32: null pointer");
    return launch_add_l1<true>(Tgt, Tpred, points, nullptr, loss_sum, dTpred, grad_scale, B, N, static_cast<hipStream_t>(stream),
                               "so3_add_l1_disentangled_f32");
}

int so3_rotate_clouds_f32(const float *P, const float *R, float *out, int transposed, int64_t B, int32_t N, void *stream) {
    SO3_CHECK_ARGS(B >= 0 && B <= (INT64_C(1) << 31) && N >= 0 && N <= 150000000, "so3_rotate_clouds_f32: B/N");
    if (B == 0 || N == 0) return 0;
    SO3_CHECK_ARGS(P != nullptr && R != nullptr && out != nullptr, "so3_rotate_clouds_f32: null pointer");
    int64_t per_wave = B / (static_cast<int64_t>(device_cus()) * 16);
    if (per_wave < 1) per_wave = 1;
    if (per_wave > 64) per_wave = 64;
    const int64_t waves = (B + per_wave - 1) / per_wave;
    const dim3 grid(static_cast<unsigned>((waves + (kBlock / 64) - 1) / (kBlock / 64))), block(kBlock);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (transposed) hipLaunchKernelGGL((k_rotate_clouds<true>), grid, block, 0, s, P, R, out, B, N, static_cast<int>(per_wave));
    else hipLaunchKernelGGL((k_rotate_clouds<false>), grid, block, 0, s, P, R, out, B, N, static_cast<int>(per_wave));
    return check_launch("so3_rotate_clouds_f32");
}

int so3_pc_normalize_f32(const float *P, float *out, float *centroid, float *scale, int64_t B, int32_t N, void *stream) {
    SO3_CHECK_ARGS(B >= 0 && B <= (INT64_C(1) << 31) && N >= 1 && N <= 150000000, "so3_pc_normalize_f32: B/N");
    if (B == 0) return 0;
    SO3_CHECK_ARGS(P != nullptr && out != nullptr, "so3_pc_normalize_f32: null pointer");
    int64_t per_wave = B / (static_cast<int64_t>(device_cus()) * 16);
    if (per_wave < 1) per_wave = 1;
    if (per_wave > 64) per_wave = 64;
    const int64_t waves = (B + per_wave - 1) / per_wave;
    const dim3 grid(static_cast<unsigned>((waves + (kBlock / 64) - 1) / (kBlock / 64))), block(kBlock);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int pw = static_cast<int>(per_wave);
    if (N <= 256) hipLaunchKernelGGL((k_pc_normalize<4>), grid, block, 0, s, P, out, centroid, scale, B, N, pw);
    else if (N <= 1024) hipLaunchKernelGGL((k_pc_normalize<16>), grid, block, 0, s, P, out, centroid, scale, B, N, pw);
    else hipLaunchKernelGGL((k_pc_normalize<0>), grid, block, 0, s, P, out, centroid, scale, B, N, pw);
    return check_launch("so3_pc_normalize_f32");
}

int so3_project_fwd_f64(const double *M, double *R, uint8_t *flip, int64_t B, void *stream) {
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B, "so3_project_fwd_f64: B");
    if (B == 0) return 0;
    SO3_CHECK_ARGS(M != nullptr && R != nullptr, "so3_project_fwd_f64: null pointer");
    hipLaunchKernelGGL((k_project_f64<false>), dim3(grid_for(B)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), M, nullptr, R, flip, B);
    return check_launch("so3_project_fwd_f64");
}

int so3_project_bwd_f64(const double *M, const double *G, double *dM, int64_t B, void *stream) {
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B, "so3_project_bwd_f64: B");
    if (B == 0) return 0;
    SO3_CHECK_ARGS(M != nullptr && G != nullptr && dM != nullptr, "so3_project_bwd_f64: null pointer");
    hipLaunchKernelGGL((k_project_f64<true>), dim3(grid_for(B)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), M, G, dM, nullptr, B);
    return check_launch("so3_project_bwd_f64");
}

size_t so3_angle_stats_workspace_bytes(void) { return sizeof(StatWork); }

int so3_angle_stats(const double *deg, const int32_t *cls, int32_t ncls, double *stats, void *workspace, int64_t B, void *stream) {
    SO3_CHECK_ARGS(B >= 0 && ncls >= 1 && ncls <= kMaxClasses, "so3_angle_stats: B / ncls (1..64)");
    SO3_CHECK_ARGS(stats != nullptr && workspace != nullptr && (B == 0 || deg != nullptr), "so3_angle_stats: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    StatWork *w = static_cast<StatWork *>(workspace);
    // (the sums, the overflow flag and the histograms of the classes in use; a memset node of this size costs two fill kernels, ~5 us each)
    // 16-byte loads of deg and 8-byte loads of cls from row 0 (mode 0) or row 1 (mode 1) on, wherever both arrays are aligned there
    auto vec_ok = [&](int64_t head) {
        return (reinterpret_cast<uintptr_t>(deg + head) & 15u) == 0 && (cls == nullptr || (reinterpret_cast<uintptr_t>(cls + head) & 7u) == 0);
    };
    const int mode = vec_ok(0) ? 0 : (vec_ok(1) ? 1 : 2);
    // one 1024-thread workgroup per CU, two rows per thread and trip (two per CU measured slower: twice the flushes and LDS histograms)
    // (SO3_STAT_GRID_MULT > 1, test builds only: more workgroups than CUs, so that the later ones START when the first ones have finished --
    // with SO3_STAT_SPINS=0 the launch-mates of every finishing workgroup then arrive after it has cleared its class)
    int64_t cap = static_cast<int64_t>(device_cus()) * SO3_STAT_GRID_MULT;
    if (cap > kStatMaxWgs) cap = kStatMaxWgs;
    const int64_t want = (B / 2 + kStatBlock - 1) / kStatBlock;
    const unsigned grid = static_cast<unsigned>(want < 1 ? 1 : (want < cap ? want : cap));
    if (ncls <= kFineClasses) {                          // few classes: bins of 1/64 octave
        k_stats_window<kFineClasses, true><<<grid, kStatBlock, 0, s>>>(deg, cls, ncls, w, B, mode);
        k_stats_collect<true><<<grid, kStatBlock, 0, s>>>(deg, cls, ncls, w, B, mode, stats);
    } else {
        if (ncls <= kStatLdsClasses) k_stats_window<kStatLdsClasses, false><<<grid, kStatBlock, 0, s>>>(deg, cls, ncls, w, B, mode);
        else k_stats_window<kMaxClasses, false><<<grid, kStatBlock, 0, s>>>(deg, cls, ncls, w, B, mode);
        k_stats_collect<false><<<grid, kStatBlock, 0, s>>>(deg, cls, ncls, w, B, mode, stats);
    }
    return check_launch("so3_angle_stats");
}

int so3_se3_update_f32(const float *out12, const float *Tinit, float *Tpred, float fx, float fy, int64_t B, void *stream) {
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B && fx != 0.f && fy != 0.f, "so3_se3_update_f32: B / fx / fy");
    if (B == 0) return 0;
    SO3_CHECK_ARGS(out12 != nullptr && Tinit != nullptr && Tpred != nullptr, "so3_se3_update_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t nunits = stream_units(B, {out12, Tinit, Tpred});
    if (nunits > 0) { so3::OpSe3Update op; op.in0 = out12; op.in1 = Tinit; op.out0 = Tpred; op.inv_fx = 1.f / fx; op.inv_fy = 1.f / fy; launch_rows<2, 2, 256>(op, nunits, s); }
    const int64_t done = nunits * so3::kUnitRows, rest = B - done;
    if (rest > 0) hipLaunchKernelGGL((k_se3_rows<false>), dim3(grid_for(rest)), dim3(kBlock), 0, s, out12 + done * 12, Tinit + done * 16, nullptr, Tpred + done * 16, 1.f / fx, 1.f / fy, rest);
    return check_launch("so3_se3_update_f32");
}

int so3_se3_update_bwd_f32(const float *out12, const float *Tinit, const float *G, float *dout12, float fx, float fy, int64_t B, void *stream) {
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B && fx != 0.f && fy != 0.f, "so3_se3_update_bwd_f32: B / fx / fy");
    if (B == 0) return 0;
    SO3_CHECK_ARGS(out12 != nullptr && Tinit != nullptr && G != nullptr && dout12 != nullptr, "so3_se3_update_bwd_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t nunits = stream_units(B, {out12, Tinit, G, dout12});
    if (nunits > 0) { so3::OpSe3UpdateBwd op; op.in0 = out12; op.in1 = Tinit; op.in2 = G; op.out0 = dout12; op.inv_fx = 1.f / fx; op.inv_fy = 1.f / fy; launch_rows<1, 3, 256>(op, nunits, s); }
    const int64_t done = nunits * so3::kUnitRows, rest = B - done;
    if (rest > 0) hipLaunchKernelGGL((k_se3_rows<true>), dim3(grid_for(rest)), dim3(kBlock), 0, s, out12 + done * 12, Tinit + done * 16, G + done * 16, dout12 + done * 12, 1.f / fx, 1.f / fy, rest);
    return check_launch("so3_se3_update_bwd_f32");
}

int so3_rotations_axis_angle_f32(const float *theta, const float *axis, float *R, int64_t B, void *stream) {
    SO3_CHECK_ARGS(B >= 0 && B <= SO3_MAX_B, "so3_rotations_axis_angle_f32: B");
    if (B == 0) return 0;
    SO3_CHECK_ARGS(theta != nullptr && axis != nullptr && R != nullptr, "so3_rotations_axis_angle_f32: null pointer");
    hipLaunchKernelGGL(k_rotations_axis_angle, dim3(grid_for(B)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), theta, axis, R, B);
    return check_launch("so3_rotations_axis_angle_f32");
}

int so3_kabsch_synth_f32(const float *P, const float *Rgt, float sigma, uint32_t seed, float *R, float *H, int64_t B, int32_t N,
                         void *stream) {
    SO3_CHECK_ARGS(B >= 0 && B <= (INT64_C(1) << 31) && N >= 0 && N <= 150000000, "so3_kabsch_synth_f32: B/N");
    if (B == 0) return 0;
    SO3_CHECK_ARGS(R != nullptr && Rgt != nullptr && (N == 0 || P != nullptr), "so3_kabsch_synth_f32: null pointer");
    int64_t cpw = B / (static_cast<int64_t>(device_cus()) * 16);
    if (cpw < 1) cpw = 1;
    if (cpw > 64) cpw = 64;
    const int64_t waves = (B + cpw - 1) / cpw;
    const int64_t blocks = (waves + (kBlock / 64) - 1) / (kBlock / 64);
    hipLaunchKernelGGL((sigma != 0.f ? k_kabsch_synth<true> : k_kabsch_synth<false>), dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), P, Rgt, sigma,
                       seed, R, H, B, N, static_cast<int>(cpw));
    return check_launch("so3_kabsch_synth_f32");
}

int so3_kabsch_f32(const float *P, const float *Q, float *R, float *H, int64_t B, int32_t N, void *stream) {
    SO3_CHECK_ARGS(B >= 0 && B <= (INT64_C(1) << 40) && N >= 0 && N <= 150000000, "so3_kabsch_f32: B/N");
    if (B == 0) return 0;
    SO3_CHECK_ARGS(R != nullptr && (N == 0 || (P != nullptr && Q != nullptr)), "so3_kabsch_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // enough waves to fill 256 CUs x 16 waves, at most 64 clouds per wave (one per lane for the SVD)
    int64_t cpw = B / (static_cast<int64_t>(device_cus()) * 16);
    if (cpw < 1) cpw = 1;
    if (cpw > 64) cpw = 64;
    const int64_t waves = (B + cpw - 1) / cpw;
    const int64_t blocks = (waves + (kBlock / 64) - 1) / (kBlock / 64);
    SO3_CHECK_ARGS(blocks <= 2147483647, "so3_kabsch_f32: B too large");
    hipLaunchKernelGGL(k_kabsch, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, s, P, Q, R, H, B, N, static_cast<int>(cpw));
    return check_launch("so3_kabsch_f32");
}

}  // extern "C"
