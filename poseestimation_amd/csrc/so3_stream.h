// so3_stream.h -- the streaming (persistent-wave) form of K1, shared by libso3proj.so and the
// in-kernel timing tool tools/ubench/k1_anatomy.hip (which instantiates it with STAMP = true).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "so3_device.h"

namespace so3 {

constexpr int kStreamBlock = 256;

// ---- K1, streaming form (f32, 16-byte aligned) ------------------------------------------------------
// The batch is cut into UNITS of 64 rows (2304 B = 144 float4).  Persistent waves, no workgroup
// barrier: in one round a wave takes NPL consecutive units (lane l owns row l of each unit), and wave w
// takes rounds w, w+W, w+2W, ...  A unit travels
//     global --dwordx4--> VGPR --ds_write_b128--> LDS --ds_read_b32 (stride 9)--> lane
// and back the same way.  The NEXT round's global loads are issued before the current round's Jacobi
// sweeps, so HBM latency hides behind the VALU work of the same wave; LDS is private to the wave and a
// wave's DS operations complete in issue order, so no s_barrier is needed.
// NPL = 2 packs two independent matrices into the halves of v_pk_* operands (so3_device.h explains why:
// dependent single-matrix chains issue at about half the VALU rate).  If the unit count is odd the last
// round's second unit is a clamped re-read whose stores are skipped (a wave-uniform branch).
constexpr int kWavesPerBlock = kStreamBlock / 64;
constexpr int kUnitRows = 64;
constexpr int kUnitFloats = kUnitRows * 9;     // 576
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Unit I/O goes through raw buffer instructions with a PER-UNIT descriptor (base = unit, num_records =
// 2304 B, or 0 for the one non-existent unit of an odd tail): float4 #lane+128 exists only on lanes < 16,
// and the hardware range check drops the other lanes' load/store instead of an exec-masked branch.
// Branch-free I/O keeps the wave's vmcnt bookkeeping exact: the compiler can wait for the prefetched
// loads with vmcnt(#younger stores) instead of draining the stores of the previous round.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kRsrcFlags = 0x00020000;           // gfx9 raw buffer, 32-bit data format
#ifndef SO3_STREAM_CPOL
#define SO3_STREAM_CPOL 2                        // cache policy of the streamed loads/stores: 2 = nt (non-temporal)
#endif
constexpr int kStreamCpol = SO3_STREAM_CPOL;     // every byte is touched once: nt keeps it from displacing L2/MALL lines
                                                 // (a plain 36 MB -> 36 MB copy: 14.8 us default policy, 13.1 us nt)

__device__ __forceinline__ rsrc_t unit_rsrc(const float *base, int64_t unit, bool exists) {
    float *p = const_cast<float *>(base) + unit * kUnitFloats;     // `unit` is wave-uniform (SGPR) by construction
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, exists ? kUnitFloats * 4 : 0, kRsrcFlags);
}
__device__ __forceinline__ void unit_fetch(f32x4 (&v)[3], rsrc_t rs, int lane) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rs, (lane + 64 * j) * 16, 0, kStreamCpol);
        v[j] = __builtin_bit_cast(f32x4, raw);
    }
}
__device__ __forceinline__ void unit_store(rsrc_t rs, const f32x4 (&v)[3], int lane) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[j]), rs, (lane + 64 * j) * 16, 0, kStreamCpol);
}
// LDS side: a unit occupies a slot of 192 float4 (3072 B): the 144 real ones plus padding, so that every
// lane can write/read float4 #lane+128 without a branch or a select (lanes >= 16 touch only the padding:
// zeros from their range-checked load on the way in, garbage that the range-checked store drops on the way out).
constexpr int kUnitSlotFloats = 192 * 4;
__device__ __forceinline__ void unit_to_lds(float *tile, const f32x4 (&v)[3], int lane) {
    f32x4 *t4 = reinterpret_cast<f32x4 *>(tile);
    t4[lane] = v[0];
    t4[lane + 64] = v[1];
    t4[lane + 128] = v[2];
}
__device__ __forceinline__ void unit_from_lds(f32x4 (&v)[3], const float *tile, int lane) {
    const f32x4 *t4 = reinterpret_cast<const f32x4 *>(tile);
    v[0] = t4[lane];
    v[1] = t4[lane + 64];
    v[2] = t4[lane + 128];
}

template <int NPL> struct LaneT;
template <> struct LaneT<1> { typedef float type; };
template <> struct LaneT<2> { typedef f32x2 type; };

// Full units only (nunits = B / 64); the host sends the < 64-row remainder to the block-tile kernel.
// WPS = resident waves per SIMD the register budget is sized for (the host launches 256*WPS blocks).
// STAMP (diagnostic builds only): per wave {s_memrealtime at entry, at exit, s_memtime at entry, (cycles | XCC<<28 | HW_ID<<32)}
// go to `stamps`, a buffer nothing else reads.
template <int NPL, bool FLIP, int WPS, bool STAMP = false, int SWEEPS = kSweeps, bool ADAPT = true>
__global__ __launch_bounds__(kStreamBlock) __attribute__((amdgpu_waves_per_eu(WPS, WPS)))
void k_project_fwd_stream(const float *__restrict__ M, float *__restrict__ R, uint8_t *__restrict__ flip, int64_t nunits,
                          unsigned long long *__restrict__ stamps) {
    unsigned long long t_real0 = 0, t_mem0 = 0, stall_cycles = 0, first_wait = 0;
    if (STAMP) { t_real0 = __builtin_amdgcn_s_memrealtime(); t_mem0 = __builtin_amdgcn_s_memtime(); }
    typedef typename LaneT<NPL>::type T;
    typedef so3::Tr<T> Tr;
    __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][NPL][kUnitSlotFloats];
    const int lane = threadIdx.x & 63;
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // SGPR: unit indices stay scalar
    float(*tile)[kUnitSlotFloats] = lds[wave_in_block];
    const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
    const int64_t nrounds = (nunits + NPL - 1) / NPL;
    int64_t t = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block;
    if (t >= nrounds) return;
    const int64_t wave_id = t;
    f32x4 in[NPL][3];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int64_t u = t * NPL + k;                      // a non-existent unit re-reads the round's first one
        unit_fetch(in[k], unit_rsrc(M, u < nunits ? u : t * NPL, true), lane);
    }
    if (STAMP) { __builtin_amdgcn_s_waitcnt(0); first_wait = __builtin_amdgcn_s_memtime() - t_mem0; }
#pragma unroll
    for (int k = 0; k < NPL; ++k) unit_to_lds(tile[k], in[k], lane);
    while (true) {
        wave_lds_fence();
        T m[9], r[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
#pragma unroll
            for (int k = 0; k < NPL; ++k) Tr::set(m[i], k, tile[k][lane * 9 + i]);
        }
        wave_lds_fence();
        const int64_t tn = t + nwaves;
        const bool more = tn < nrounds;
        const int64_t tf = more ? tn : t;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {                     // prefetch: in flight during the sweeps below.
            const int64_t u = tf * NPL + k;                 // After the last round the descriptor is empty: the
            unit_fetch(in[k], unit_rsrc(M, u < nunits ? u : tf * NPL, more), lane);   // loads return 0, no traffic
        }
        if constexpr (SWEEPS < 0) {                         // diagnostic: pure data movement, no arithmetic
#pragma unroll
            for (int i = 0; i < 9; ++i) r[i] = m[i];
        } else {
            const auto f = signed_svd<false, T, SWEEPS, ADAPT>(m);
            rotation_from(f, r);
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
#pragma unroll
            for (int k = 0; k < NPL; ++k) tile[k][lane * 9 + i] = Tr::get(r[i], k);
        }
        wave_lds_fence();
        f32x4 o[NPL][3];
#pragma unroll
        for (int k = 0; k < NPL; ++k) unit_from_lds(o[k], tile[k], lane);
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int64_t u = t * NPL + k;
            const bool exists = u < nunits;                 // wave-uniform; a non-existent unit's stores are dropped
            unit_store(unit_rsrc(R, exists ? u : 0, exists), o[k], lane);
            if (FLIP) {
                float mk_[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) mk_[i] = Tr::get(m[i], k);
                if (exists) flip[u * kUnitRows + lane] = det_negative(mk_) ? 1 : 0;
            }
        }
        if (!more) break;
        // The prefetched units land in LDS here, at the END of the body: the loads are older than this
        // round's stores, so the wait the compiler places is vmcnt(#stores), not a drain of the stores.
        unsigned long long w0 = 0;
        if (STAMP) { __builtin_amdgcn_sched_barrier(0); w0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0x0F70 | 6); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int k = 0; k < NPL; ++k) unit_to_lds(tile[k], in[k], lane);
        if (STAMP) { __builtin_amdgcn_sched_barrier(0); stall_cycles += __builtin_amdgcn_s_memtime() - w0; }
        t = tn;
    }
    if (STAMP) {
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) {
            stamps[6 * wave_id + 0] = t_real0;
            stamps[6 * wave_id + 1] = __builtin_amdgcn_s_memrealtime();
            stamps[6 * wave_id + 2] = t_mem0;
            stamps[6 * wave_id + 4] = stall_cycles;          // cycles spent waiting for prefetched units
            stamps[6 * wave_id + 5] = first_wait;            // cycles from wave start until its first unit arrived
            // HW_REG_HW_ID (id 4) and HW_REG_XCC_ID (id 20): which XCD / SE / CU / SIMD ran this wave
            const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
            const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
            // cycles in bits 0..27, XCC id in 28..31, HW_ID in 32..63
            stamps[6 * wave_id + 3] = ((__builtin_amdgcn_s_memtime() - t_mem0) & 0xFFFFFFFull)
                                      | (static_cast<unsigned long long>(xcc & 0xF) << 28) | (static_cast<unsigned long long>(hw) << 32);
        }
    }
}


// ---- K4, streaming form ---------------------------------------------------------------------------------
// theta = acos(clamp((tr(R1^T R2) - 1)/2)) in float64 on float32 data (rotation_representation.py:230-242).
// 512-thread workgroups (8 waves, three per CU: 48 KB of LDS each), each wave walks units w, w+W, ... with the same
// wave-private LDS staging as K1 and keeps a per-lane float64 partial sum; ONE atomicAdd per workgroup then
// carries the fused (sum) reduction -- same-address float64 atomics cost ~12 ns each, so a per-tile atomic
// (thousands per launch) would dominate the kernel.
constexpr int kAngleBlock = 512;
constexpr int kAngleWaves = kAngleBlock / 64;

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <bool WANT_DEG, bool WANT_SUM>
__global__ __launch_bounds__(kAngleBlock) void k_angle_error_stream(const float *__restrict__ R1, const float *__restrict__ R2,
                                                                    double *__restrict__ out, double *__restrict__ sum_count,
                                                                    int32_t *__restrict__ range_flag, double unit_scale,
                                                                    int64_t nunits) {
    __shared__ __attribute__((aligned(16))) float lds[kAngleWaves][2][kUnitSlotFloats];
    __shared__ double red[kAngleWaves];
    const int lane = threadIdx.x & 63;
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float(*tile)[kUnitSlotFloats] = lds[wave_in_block];
    const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kAngleWaves;
    double acc = 0.0;
    bool any_bad = false;
    int64_t t = static_cast<int64_t>(blockIdx.x) * kAngleWaves + wave_in_block;
    if (t < nunits) {
        f32x4 in[2][3];
        unit_fetch(in[0], unit_rsrc(R1, t, true), lane);
        unit_fetch(in[1], unit_rsrc(R2, t, true), lane);
        while (true) {
            unit_to_lds(tile[0], in[0], lane);
            unit_to_lds(tile[1], in[1], lane);
            wave_lds_fence();
            float a[9], b[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) { a[i] = tile[0][lane * 9 + i]; b[i] = tile[1][lane * 9 + i]; }
            wave_lds_fence();
            const int64_t tn = t + nwaves;
            const bool more = tn < nunits;
            unit_fetch(in[0], unit_rsrc(R1, more ? tn : t, more), lane);      // prefetch (empty descriptor at the end)
            unit_fetch(in[1], unit_rsrc(R2, more ? tn : t, more), lane);
            double tr = 0.0;                                 // tr(R1^T R2) = sum_ij R1_ij R2_ij, float64
#pragma unroll
            for (int i = 0; i < 9; ++i) tr = fma(static_cast<double>(a[i]), static_cast<double>(b[i]), tr);
            const double c_raw = (tr - 1.0) * 0.5;
            any_bad |= (c_raw < -1.1 || c_raw > 1.1);        // NaN compares false, as torch.any(...) does
            double c = fmin(fmax(c_raw, -1.0), 1.0);         // torch.clamp ...
            if (c_raw != c_raw) c = c_raw;                   // ... which keeps NaN (fmin/fmax drop it)
            const double ang = acos(c) * unit_scale;
            if (WANT_DEG) out[t * kUnitRows + lane] = ang;
            if (WANT_SUM) acc += ang;
            if (!more) break;
            t = tn;
        }
    }
    if (range_flag != nullptr && __any(any_bad)) {
        if (lane == 0) atomicOr(range_flag, 1);
    }
    if (WANT_SUM) {
        acc = wave_sum_f64(acc);
        if (lane == 0) red[wave_in_block] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double total = 0.0;
#pragma unroll
            for (int w = 0; w < kAngleWaves; ++w) total += red[w];
            atomicAdd(sum_count, total);
        }
    }
}

}  // namespace so3
