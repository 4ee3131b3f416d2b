// Microbenchmark: per-SIMD issue rates of the VALU instructions the Jacobi kernel is made of.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip ; run on an MI355X.
// Prints wave-instructions per cycle per CU (4 SIMDs) at the measured average clock estimate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float float2v __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4096;
constexpr int ACC = 8;

template <int MODE>
__global__ __launch_bounds__(256) void bench(float *out, float a, float b) {
    float x[ACC];
    float2v y[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = float2v{x[i], x[i] + 0.5f}; }
    const float2v a2 = {a, a * 1.0001f}, b2 = {b, b * 0.9999f};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) {
            if (MODE == 0) x[i] = fmaf(x[i], a, b);                                   // v_fma_f32
            if (MODE == 1) y[i] = __builtin_elementwise_fma(y[i], a2, b2);            // v_pk_fma_f32
            if (MODE == 2) x[i] = __builtin_amdgcn_rsqf(x[i]) + 1.5f;                 // v_rsq_f32 + v_add
            if (MODE == 3) x[i] = __builtin_amdgcn_sqrtf(x[i]) + 1.5f;                // v_sqrt_f32 + v_add
            if (MODE == 4) x[i] = (x[i] > a) ? x[i] * b : x[i] + a;                   // cmp + cndmask + mul + add
            if (MODE == 5) y[i] = y[i] * a2;                                          // v_pk_mul_f32
            if (MODE == 6) x[i] = x[i] * a;                                           // v_mul_f32
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += x[i] + y[i].x + y[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
void run(const char *name, int ops_per_inner, float *d) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.0001f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.0001f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    // wave-instructions: blocks*4 waves * ITERS*ACC*ops
    const double winst = (double)blocks * 4 * ITERS * ACC * ops_per_inner;
    const double per_cu_per_s = winst / 256.0 / (ms * 1e-3);
    printf("%-28s %8.3f ms   %.3f wave-instr/ns/CU  (= %.2f per cycle per CU at 2.4 GHz; %.2f cycles per wave-instr per SIMD)\n",
           name, ms, per_cu_per_s * 1e-9, per_cu_per_s / 2.4e9, 4.0 * 2.4e9 / per_cu_per_s);
}

int main() {
    float *d; CHECK(hipMalloc(&d, 4));
    run<0>("v_fma_f32", 1, d);
    run<1>("v_pk_fma_f32", 1, d);
    run<2>("v_rsq_f32+v_add", 2, d);
    run<3>("v_sqrt_f32+v_add", 2, d);
    run<4>("cmp+cndmask+mul+add", 4, d);
    run<5>("v_pk_mul_f32", 1, d);
    run<6>("v_mul_f32", 1, d);
    return 0;
}
