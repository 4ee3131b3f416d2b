// Microbenchmark: per-SIMD issue rates of the VALU instructions the Jacobi kernel is made of,
// as a function of instruction-level parallelism inside a wave (ACC independent chains) and
// of waves per SIMD.  Also reports the shader clock (s_memtime ticks / wall time).
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float float2v __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 8192;

template <int MODE, int ACC>
__global__ __launch_bounds__(256) void bench(float *out, unsigned long long *clk, float a, float b) {
    float x[ACC];
    float2v y[ACC];
    double z[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = float2v{x[i], x[i] + 0.5f}; z[i] = x[i]; }
    const double ad = a, bd = b;
    const float2v a2 = {a, a * 1.0001f}, b2 = {b, b * 0.9999f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) {
            if (MODE == 0) x[i] = fmaf(x[i], a, b);                                   // v_fma_f32
            if (MODE == 1) y[i] = __builtin_elementwise_fma(y[i], a2, b2);            // v_pk_fma_f32
            if (MODE == 2) x[i] = __builtin_amdgcn_rsqf(x[i]) + 1.5f;                 // v_rsq_f32 + v_add
            if (MODE == 3) x[i] = fmaf(x[i], x[i], b);                                // v_fma_f32, all-VGPR operands
            if (MODE == 4) x[i] = (x[i] > a) ? x[i] * b : x[i] + a;                   // cmp + cndmask + mul + add
            if (MODE == 5) z[i] = __builtin_fma(z[i], ad, bd);                        // v_fma_f64
            if (MODE == 6) { x[i] = x[i] + a; z[i] = z[i] + static_cast<double>(x[i]); }   // v_add_f32 + v_cvt_f64_f32 + v_add_f64
            if (MODE == 7) { x[i] = x[i] + a; z[i] = z[i] + ad; }                     // v_add_f32 + v_add_f64
            if (MODE == 8) z[i] = z[i] * ad;                                          // v_mul_f64
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += x[i] + y[i].x + y[i].y + static_cast<float>(z[i]);
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int MODE, int ACC>
void run(const char *name, int ops_per_inner, float *d, unsigned long long *clk, int blocks_per_cu) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 256 * blocks_per_cu;
    hipLaunchKernelGGL((bench<MODE, ACC>), dim3(blocks), dim3(256), 0, 0, d, clk, 1.0001f, 0.0001f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((bench<MODE, ACC>), dim3(blocks), dim3(256), 0, 0, d, clk, 1.0001f, 0.0001f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long ticks; CHECK(hipMemcpy(&ticks, clk, 8, hipMemcpyDeviceToHost));
    const double winst = (double)blocks * 4 * ITERS * ACC * ops_per_inner;
    const double per_simd_per_ns = winst / 1024.0 / (ms * 1e6);
    const double cyc_per_inst = (double)ticks / ((double)ITERS * ACC * ops_per_inner) / blocks_per_cu;   // per SIMD: waves/SIMD = blocks_per_cu
    printf("%-22s ACC=%d waves/SIMD=%d  %8.3f ms  %.3f winstr/ns/SIMD  memtime ticks/ns %.3f  ticks per winstr per SIMD %.2f\n",
           name, ACC, blocks_per_cu, ms, per_simd_per_ns, (double)ticks / (ms * 1e6), cyc_per_inst);
}

int main() {
    float *d; CHECK(hipMalloc(&d, 4));
    unsigned long long *clk; CHECK(hipMalloc(&clk, 8));
    for (int bpc : {2, 4}) {
        run<5, 1>("v_fma_f64", 1, d, clk, bpc);
        run<5, 8>("v_fma_f64", 1, d, clk, bpc);
        run<8, 8>("v_mul_f64", 1, d, clk, bpc);
        run<6, 8>("add_f32+cvt_f64_f32+add_f64", 3, d, clk, bpc);
        run<7, 8>("add_f32+add_f64", 2, d, clk, bpc);
    }
    for (int bpc : {1, 2, 4, 8}) {
        run<0, 1>("v_fma_f32", 1, d, clk, bpc);
        run<0, 2>("v_fma_f32", 1, d, clk, bpc);
        run<0, 4>("v_fma_f32", 1, d, clk, bpc);
        run<0, 8>("v_fma_f32", 1, d, clk, bpc);
    }
    for (int bpc : {4, 8}) {
        run<3, 1>("v_fma_f32 vgpr-only", 1, d, clk, bpc);
        run<3, 8>("v_fma_f32 vgpr-only", 1, d, clk, bpc);
        run<1, 1>("v_pk_fma_f32", 1, d, clk, bpc);
        run<1, 8>("v_pk_fma_f32", 1, d, clk, bpc);
        run<2, 1>("v_rsq_f32+v_add", 2, d, clk, bpc);
        run<2, 8>("v_rsq_f32+v_add", 2, d, clk, bpc);
        run<4, 1>("cmp+cndmask+mul+add", 4, d, clk, bpc);
        run<4, 8>("cmp+cndmask+mul+add", 4, d, clk, bpc);
    }
    return 0;
}
