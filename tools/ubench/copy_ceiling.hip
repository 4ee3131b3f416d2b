// copy_ceiling: what a pure 36 MB -> 36 MB streaming copy reaches on MI355X, per launch, with rotating
// buffers (so the Infinity Cache cannot serve it).  This is the data-movement ceiling for K1 at 1M rows.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int64_t N4 = 1000000LL * 9 / 4;   // float4 count: 36 MB
constexpr int NBUF = 8;

template <int NT, int UNROLL>
__global__ __launch_bounds__(256) void copy_gs(const f32x4 *__restrict__ in, f32x4 *__restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { if (NT) __builtin_nontemporal_store(v[u], out + i + u * stride); else out[i + u * stride] = v[u]; }
    }
    for (; i < n; i += stride) { f32x4 v = in[i]; out[i] = v; }
}

// each block copies one contiguous chunk (block-contiguous instead of grid-stride)
template <int NT>
__global__ __launch_bounds__(256) void copy_chunk(const f32x4 *__restrict__ in, f32x4 *__restrict__ out, int64_t n, int64_t per_block) {
    const int64_t b0 = (int64_t)blockIdx.x * per_block;
    const int64_t b1 = b0 + per_block < n ? b0 + per_block : n;
    for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) {
        f32x4 v = NT ? __builtin_nontemporal_load(in + i) : in[i];
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
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
    f32x4 *in[NBUF], *out[NBUF];
    for (int i = 0; i < NBUF; ++i) { CHECK(hipMalloc(&in[i], N4 * 16)); CHECK(hipMalloc(&out[i], N4 * 16)); CHECK(hipMemset(in[i], 1, N4 * 16)); }
    CHECK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep) {
        timeit("grid-stride 2048 blocks", [&](int i) { hipLaunchKernelGGL((copy_gs<0, 1>), dim3(2048), dim3(256), 0, 0, in[i % NBUF], out[i % NBUF], N4); });
        timeit("grid-stride 2048 blocks unroll4", [&](int i) { hipLaunchKernelGGL((copy_gs<0, 4>), dim3(2048), dim3(256), 0, 0, in[i % NBUF], out[i % NBUF], N4); });
        timeit("grid-stride 1024 blocks unroll4", [&](int i) { hipLaunchKernelGGL((copy_gs<0, 4>), dim3(1024), dim3(256), 0, 0, in[i % NBUF], out[i % NBUF], N4); });
        timeit("grid-stride 4096 blocks unroll2", [&](int i) { hipLaunchKernelGGL((copy_gs<0, 2>), dim3(4096), dim3(256), 0, 0, in[i % NBUF], out[i % NBUF], N4); });
        timeit("grid-stride 2048 blocks unroll4 nt", [&](int i) { hipLaunchKernelGGL((copy_gs<1, 4>), dim3(2048), dim3(256), 0, 0, in[i % NBUF], out[i % NBUF], N4); });
        timeit("one-shot 8790 blocks (1 float4/thread)", [&](int i) { hipLaunchKernelGGL((copy_gs<0, 1>), dim3((N4 + 255) / 256), dim3(256), 0, 0, in[i % NBUF], out[i % NBUF], N4); });
        timeit("one-shot nt", [&](int i) { hipLaunchKernelGGL((copy_gs<1, 1>), dim3((N4 + 255) / 256), dim3(256), 0, 0, in[i % NBUF], out[i % NBUF], N4); });
        timeit("chunked 2048 blocks", [&](int i) { hipLaunchKernelGGL((copy_chunk<0>), dim3(2048), dim3(256), 0, 0, in[i % NBUF], out[i % NBUF], N4, (N4 + 2047) / 2048); });
        timeit("chunked 2048 blocks nt", [&](int i) { hipLaunchKernelGGL((copy_chunk<1>), dim3(2048), dim3(256), 0, 0, in[i % NBUF], out[i % NBUF], N4, (N4 + 2047) / 2048); });
        timeit("same buffer (cache resident) gs 2048 u4", [&](int i) { hipLaunchKernelGGL((copy_gs<0, 4>), dim3(2048), dim3(256), 0, 0, in[0], out[0], N4); });
    }
    return 0;
}
