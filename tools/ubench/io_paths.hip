// io_paths: how fast can 1M 36-byte records be read into "one record per lane" registers and written
// back, per launch, by access pattern?  (The arithmetic of K1 then has to hide under the best of these.)
//   direct : each lane reads/writes its own 36 B as 3 x dwordx3 (lane stride 36 B)
//   lds    : coalesced float4 <-> LDS <-> lane (stride-9 dword LDS accesses)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
constexpr int64_t ROWS = 1000000;
constexpr int NBUF = 8;

typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t mk_rsrc(const float *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, bytes, 0x00020000);
}
template <int NT>
__device__ __forceinline__ void load9_direct(rsrc_t rs, unsigned byte_off, float (&m)[9]) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, byte_off + 12 * j, 0, NT ? 2 : 0);
        m[3 * j] = __uint_as_float(v.x); m[3 * j + 1] = __uint_as_float(v.y); m[3 * j + 2] = __uint_as_float(v.z);
    }
}
template <int NT>
__device__ __forceinline__ void store9_direct(rsrc_t rs, unsigned byte_off, const float (&m)[9]) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const u32x3 v = {__float_as_uint(m[3 * j]), __float_as_uint(m[3 * j + 1]), __float_as_uint(m[3 * j + 2])};
        __builtin_amdgcn_raw_buffer_store_b96(v, rs, byte_off + 12 * j, 0, NT ? 2 : 0);
    }
}

// one record per lane, one-shot grid (ROWS/256 blocks)
template <int IN_LDS, int OUT_LDS, int NT>
__global__ __launch_bounds__(256) void io_oneshot(const float *__restrict__ M, float *__restrict__ R, int64_t rows) {
    __shared__ __attribute__((aligned(16))) float lds[4][576];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t unit = (int64_t)blockIdx.x * 4 + w;
    if (unit * 64 >= rows) return;
    float m[9];
    const float *src = M + unit * 576;
    float *dst = R + unit * 576;
    f32x4 *t4 = reinterpret_cast<f32x4 *>(lds[w]);
    if (IN_LDS) {
        const f32x4 *s4 = reinterpret_cast<const f32x4 *>(src);
        f32x4 a = NT ? __builtin_nontemporal_load(s4 + lane) : s4[lane];
        f32x4 b = NT ? __builtin_nontemporal_load(s4 + lane + 64) : s4[lane + 64];
        f32x4 c = {0, 0, 0, 0};
        if (lane < 16) c = NT ? __builtin_nontemporal_load(s4 + lane + 128) : s4[lane + 128];
        t4[lane] = a; t4[lane + 64] = b; if (lane < 16) t4[lane + 128] = c;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 9; ++i) m[i] = lds[w][lane * 9 + i];
    } else {
        load9_direct<NT>(mk_rsrc(M, (unsigned)(rows * 36)), (unsigned)(unit * 2304 + lane * 36), m);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = m[i] * 1.0001f + 0.5f;
    if (OUT_LDS) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 9; ++i) lds[w][lane * 9 + i] = m[i];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        f32x4 *d4 = reinterpret_cast<f32x4 *>(dst);
        f32x4 a = t4[lane], b = t4[lane + 64];
        if (NT) { __builtin_nontemporal_store(a, d4 + lane); __builtin_nontemporal_store(b, d4 + lane + 64); } else { d4[lane] = a; d4[lane + 64] = b; }
        if (lane < 16) { f32x4 c = t4[lane + 128]; if (NT) __builtin_nontemporal_store(c, d4 + lane + 128); else d4[lane + 128] = c; }
    } else {
        store9_direct<NT>(mk_rsrc(R, (unsigned)(rows * 36)), (unsigned)(unit * 2304 + lane * 36), m);
    }
}

template <class F> void timeit(const char *name, F launch) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch(i);
    CHECK(hipDeviceSynchronize());
    const int K = 40;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < K; ++i) launch(i);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / K;
    printf("%-44s %7.2f us/launch  %6.0f GB/s  (%.1f%% of 8 TB/s)\n", name, us, 72e6 / us * 1e-3, 72e6 / us * 1e-3 / 80.0);
}

int main() {
    float *in[NBUF], *out[NBUF];
    for (int i = 0; i < NBUF; ++i) { CHECK(hipMalloc(&in[i], ROWS * 36)); CHECK(hipMalloc(&out[i], ROWS * 36)); CHECK(hipMemset(in[i], 0, ROWS * 36)); }
    CHECK(hipDeviceSynchronize());
    const unsigned g = (unsigned)((ROWS / 64 + 3) / 4);
#define RUN(A, B, C, NAME) timeit(NAME, [&](int i) { hipLaunchKernelGGL((io_oneshot<A, B, C>), dim3(g), dim3(256), 0, 0, in[i % NBUF], out[i % NBUF], ROWS); })
    for (int rep = 0; rep < 2; ++rep) {
        RUN(1, 1, 0, "lds in, lds out");
        RUN(1, 1, 1, "lds in, lds out, nt");
        RUN(0, 0, 0, "direct in, direct out");
        RUN(0, 0, 1, "direct in, direct out, nt");
        RUN(0, 1, 0, "direct in, lds out");
        RUN(0, 1, 1, "direct in, lds out, nt");
        RUN(1, 0, 0, "lds in, direct out");
        RUN(1, 0, 1, "lds in, direct out, nt");
    }
    return 0;
}
