// k1_anatomy: stand-alone timing / in-kernel-stamp tool for the streaming K1 kernel.
// Instantiates the SAME kernel template the library ships (csrc/so3_rows.h) -- K1 and a copy through the engine --
// times each with hipEvents over rotating buffers, and (STAMP build of the same
// template) reports per-wave lifetimes and the shader clock from s_memtime / s_memrealtime.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -o k1_anatomy k1_anatomy.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "../../poseestimation_amd/csrc/so3_rows.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int64_t ROWS = 1000000;
constexpr int NBUF = 8;
static int g_launches = 40;
static float *g_rot[NBUF];          // rotations: the second input of the two-input operations ("two" mode)
static double *g_acc;
static int *g_flag;

__global__ void fill(float *p, int64_t n, unsigned seed) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        unsigned h2 = h * 747796405u + 2891336453u; h2 ^= h2 >> 16; h2 *= 2246822519u; h2 ^= h2 >> 13;
        const float u1 = ((h & 0xFFFFFF) + 1) / 16777217.0f, u2 = (h2 & 0xFFFFFF) / 16777216.0f;
        p[i] = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);  // N(0,1) (Box-Muller): the benchmark's distribution
    }
}

// The engine moving bytes with no arithmetic: what the data path alone costs (this tool's own operation -- the product
// headers carry no copy mode).
struct OpCopy : so3::OpBase {
    static constexpr int kIn0 = 4, kIn1 = 0, kOut0 = 4, kOut1 = 0;
    static constexpr int kFixedRounds = SO3_K1_FIXED_ROUNDS;      // (-DSO3_K1_FIXED_ROUNDS=k: the non-persistent shape, as K1's)
    template <class T, int NPL>
    __device__ __forceinline__ void compute(so3::Rows<T, OpCopy> &rows, so3::RowCtx<NPL> &) const {
#pragma unroll
        for (int i = 0; i < 9; ++i) rows.o0[i] = rows.a[i];
    }
};

template <class Op, int NPL, int WPS, int BLOCK = 256>
void run(const char *what, float **in, float **out, unsigned long long *stamps_d) {
    const int64_t nunits = ROWS / 64;
    const int64_t rounds = (nunits + NPL - 1) / NPL;
    constexpr int kW = BLOCK / 64;
    const int64_t want = (rounds + kW - 1) / kW;
    unsigned blocks = (unsigned)std::min<int64_t>(want, 256LL * 4 * WPS / kW);
    if (Op::kFixedRounds > 0) blocks = (unsigned)(((rounds + Op::kFixedRounds - 1) / Op::kFixedRounds + kW - 1) / kW);   // a wave per k consecutive rounds
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // second input (operations that have one): the OTHER buffer set's outputs, i.e. rotations once main() has run K1 over them;
    // reducing operations add onto a scratch accumulator with one atomic per workgroup (the library's no-workspace path)
    auto mk = [&](int i) {
        Op op; op.in0 = in[i % NBUF]; op.out0 = out[i % NBUF];
        if constexpr (Op::kIn1 != 0) op.in1 = g_rot[(i + 1) % NBUF];
        if constexpr (Op::kOut0 == 0) op.out0 = nullptr;
        if constexpr (Op::kReduce) { op.sum_count = g_acc; op.range_flag = g_flag; op.unit_scale = 57.29577951308232; op.count = (double)ROWS; }
        return op;
    };
    for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL((so3::k_rows<Op, NPL, WPS, BLOCK, false>), dim3(blocks), dim3(BLOCK), 0, 0, mk(i), nunits, nullptr);
    CHECK(hipDeviceSynchronize());
    const int K = g_launches;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < K; ++i)
        hipLaunchKernelGGL((so3::k_rows<Op, NPL, WPS, BLOCK, false>), dim3(blocks), dim3(BLOCK), 0, 0, mk(i), nunits, nullptr);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / K;
    // stamped build of the same template: wave lifetimes and clock
    CHECK(hipMemset(stamps_d, 0, 8 * 46 * 8192));
    hipLaunchKernelGGL((so3::k_rows<Op, NPL, WPS, BLOCK, true>), dim3(blocks), dim3(BLOCK), 0, 0, mk(0), nunits, stamps_d);
    CHECK(hipDeviceSynchronize());
    const int64_t nw = Op::kFixedRounds > 0 ? (rounds + Op::kFixedRounds - 1) / Op::kFixedRounds : std::min<int64_t>((int64_t)blocks * kW, rounds);
    std::vector<unsigned long long> st6(6 * nw);
    CHECK(hipMemcpy(st6.data(), stamps_d, st6.size() * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> rs(40 * nw);
    CHECK(hipMemcpy(rs.data(), stamps_d + 6 * (int64_t)blocks * kW, rs.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long r0 = ~0ull, r1 = 0; double life = 0, clk = 0; std::vector<double> starts;
    for (int64_t w = 0; w < nw; ++w) {
        r0 = std::min(r0, st6[6 * w]); r1 = std::max(r1, st6[6 * w + 1]);
        life += (double)(st6[6 * w + 1] - st6[6 * w]);
        clk += (double)(st6[6 * w + 3] & 0xFFFFFFFull) / (double)(st6[6 * w + 1] - st6[6 * w]);
    }
    for (int64_t w = 0; w < nw; ++w) starts.push_back((double)(st6[6 * w] - r0) * 0.01);
    std::sort(starts.begin(), starts.end());
    {   // per-wave dump for offline analysis (docs/history/tools.tar.gz:tools/wave_csv.py, phase_csv.py)
        char name[128]; snprintf(name, sizeof name, "gpurun_out/k1_waves_%s_npl%d_wps%d_fixed%d.csv", what, NPL, WPS, Op::kFixedRounds);
        FILE *fh = fopen(name, "w");
        if (fh) {
            fprintf(fh, "wave,rounds,start_us,end_us,cycles,hw_id,xcc,first_wait_cyc,phases\n");
            for (int64_t w = 0; w < nw; ++w) {
                const int64_t nr = (int64_t)(st6[6 * w + 4] >> 48);
                fprintf(fh, "%lld,%lld,%.2f,%.2f,%llu,%llu,%llu,%llu", (long long)w, (long long)nr, (double)(st6[6 * w] - r0) * 0.01,
                        (double)(st6[6 * w + 1] - r0) * 0.01, st6[6 * w + 3] & 0xFFFFFFFull, st6[6 * w + 3] >> 32, (st6[6 * w + 3] >> 28) & 0xF, st6[6 * w + 5]);
                for (int j = 0; j < 40; ++j) fprintf(fh, "%s%.2f", j ? " " : ",", rs[40 * w + j] ? (double)(rs[40 * w + j] - r0) * 0.01 : -1.0);
                fprintf(fh, "\n");
            }
            fclose(fh);
        }
    }
    printf("%-8s NPL=%d WPS=%d block=%d fixed=%d blocks=%u : %.2f us/launch (%.0f GB/s, %.1f%% of 8 TB/s) | stamped: span %.2f us, mean wave life %.2f us, "
           "wave start p50 %.2f p99 %.2f max %.2f us, memtime/realtime %.3f (x100 MHz)\n",
           what, NPL, WPS, BLOCK, Op::kFixedRounds, blocks, us, 72.0 * ROWS / us * 1e-3, 72.0 * ROWS / us * 1e-3 / 80.0, (double)(r1 - r0) * 0.01, life / nw * 0.01,
           starts[nw / 2], starts[(size_t)(nw * 0.99)], starts.back(), clk / nw);
    {   // the waves' first four rounds: arithmetic per round, period between the starts of consecutive rounds, what lies in front of the
        // first round and behind the last stamped one (wall clock, 10 ns resolution)
        double arith = 0, period = 0, fill = 0, tail = 0; long long na = 0, np_ = 0, nf = 0, nt = 0;
        for (int64_t w = 0; w < nw; ++w) {
            const int64_t nr = std::min<int64_t>((int64_t)(st6[6 * w + 4] >> 48), 4);
            for (int64_t r = 0; r < nr; ++r) {
                const unsigned long long b = rs[40 * w + 10 * r + 3], e = rs[40 * w + 10 * r + 4];
                if (b && e) { arith += (double)(e - b); ++na; }
                if (r + 1 < nr && b && rs[40 * w + 10 * (r + 1) + 3]) { period += (double)(rs[40 * w + 10 * (r + 1) + 3] - b); ++np_; }
            }
            if (nr > 0 && rs[40 * w + 3]) { fill += (double)(rs[40 * w + 3] - st6[6 * w]); ++nf; }
            if (nr > 0 && nr < 4 && rs[40 * w + 10 * (nr - 1) + 4]) { tail += (double)(st6[6 * w + 1] - rs[40 * w + 10 * (nr - 1) + 4]); ++nt; }
        }
        printf("         per wave: start -> first round's arithmetic %.2f us | arithmetic of a round %.2f us | start of a round -> start of the next %.2f us | "
               "end of the last round's arithmetic -> wave end %.2f us (waves with < 4 rounds)\n",
               nf ? fill / nf * 0.01 : 0.0, na ? arith / na * 0.01 : 0.0, np_ ? period / np_ * 0.01 : 0.0, nt ? tail / nt * 0.01 : 0.0);
    }
    fflush(stdout);
}

int main(int argc, char **argv) {
    float *in[NBUF], *out[NBUF];
    for (int i = 0; i < NBUF; ++i) {
        CHECK(hipMalloc(&in[i], ROWS * 9 * 4)); CHECK(hipMalloc(&out[i], ROWS * 9 * 4));
        hipLaunchKernelGGL(fill, dim3((ROWS * 9 + 255) / 256), dim3(256), 0, 0, in[i], ROWS * 9, 1234u + i);
    }
    unsigned long long *stamps; CHECK(hipMalloc(&stamps, 8 * 46 * 8192));
    CHECK(hipDeviceSynchronize());
    g_launches = argc > 1 ? atoi(argv[1]) : 1000;
    typedef so3::OpProject<4, false> K1;
    CHECK(hipMalloc(&g_acc, 4 * sizeof(double))); CHECK(hipMemset(g_acc, 0, 4 * sizeof(double)));
    CHECK(hipMalloc(&g_flag, sizeof(int))); CHECK(hipMemset(g_flag, 0, sizeof(int)));
    if (argc > 2 && argv[2][0] == 't') {
        // "two": where a launch of the two-input kernels spends its time, beside K1 and the pure reader K4
        for (int i = 0; i < NBUF; ++i) {          // rotations for the second input: K1 over freshly drawn rows
            CHECK(hipMalloc(&g_rot[i], ROWS * 9 * 4));
            float *tmp; CHECK(hipMalloc(&tmp, ROWS * 9 * 4));
            hipLaunchKernelGGL(fill, dim3((ROWS * 9 + 255) / 256), dim3(256), 0, 0, tmp, ROWS * 9, 777u + i);
            K1 op; op.in0 = tmp; op.out0 = g_rot[i];
            hipLaunchKernelGGL((so3::k_rows<K1, 2, 3, 256, false>), dim3(768), dim3(256), 0, 0, op, ROWS / 64, nullptr);
            CHECK(hipDeviceSynchronize()); CHECK(hipFree(tmp));
        }
        for (int rep = 0; rep < 2; ++rep) {
            run<K1, 2, 3>("k1", in, out, stamps);
            run<so3::OpProjectAngle<4, false, false, true, true>, 2, 2>("k1+k4", in, out, stamps);
            run<so3::OpProjectBwd<4>, 2, 2>("k2", in, out, stamps);
            run<so3::OpAngle<false, true>, 1, 4, 1024>("k4", g_rot, out, stamps);
        }
        return 0;
    }
    if (argc > 2 && argv[2][0] == 'p') {
        // "plain": K1 with ONE matrix per lane (v_fma_f32 instead of v_pk_fma_f32: twice the instructions at half the cost each) beside the
        // shipped packed form and the copy -- does the clock a device holds under K1 depend on the packed instructions?
        for (int rep = 0; rep < 2; ++rep) {
            run<K1, 2, 3>("k1", in, out, stamps);
            run<K1, 1, 4>("k1", in, out, stamps);
            run<K1, 1, 5>("k1", in, out, stamps);
            run<OpCopy, 2, 3>("copy", in, out, stamps);
        }
        return 0;
    }
    for (int rep = 0; rep < 3; ++rep) {
        run<K1, 2, 3>("k1", in, out, stamps);
        run<OpCopy, 2, 3>("copy", in, out, stamps);
    }
    run<OpCopy, 2, 4>("copy", in, out, stamps);
    run<OpCopy, 1, 4>("copy", in, out, stamps);
    run<OpCopy, 1, 8>("copy", in, out, stamps);
    return 0;
}
