/* c_abi_demo.c -- libso3proj.so from plain C: no Python, no torch.
 *
 * Projects a batch of 3x3 matrices onto SO(3) (so3_project_fwd_f32), measures the geodesic angle to a second batch
 * (so3_angle_error_v2 with the fused (sum, count) reduction) and checks on the host that every output is a rotation.
 *
 *   gcc -std=c11 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_demo.c -Lposeestimation_amd -lso3proj \
 *       -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/poseestimation_amd -Wl,-rpath,/opt/rocm/lib -o c_abi_demo
 *   ./c_abi_demo [rows]
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "so3proj.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define SO3_OK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s: %d (%s)\n", #x, rc_, so3_last_error()); return 3; } } while (0)

static float gauss(uint64_t *s) {                       /* xorshift + Box-Muller; any Gaussian-ish input will do */
    *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17;
    const double u1 = ((*s >> 11) + 1.0) / 9007199254740993.0;
    *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17;
    const double u2 = (*s >> 11) / 9007199254740992.0;
    return (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
}

int main(int argc, char **argv) {
    const int64_t B = argc > 1 ? atoll(argv[1]) : 100003;          /* not a multiple of 64: exercises the remainder path */
    const size_t bytes = (size_t)B * 9 * sizeof(float);
    float *hM = (float *)malloc(bytes), *hT = (float *)malloc(bytes), *hR = (float *)malloc(bytes);
    uint8_t *hflip = (uint8_t *)malloc((size_t)B);
    uint64_t seed = 0x9E3779B97F4A7C15ull;
    for (int64_t i = 0; i < B * 9; ++i) { hM[i] = gauss(&seed); hT[i] = gauss(&seed); }

    float *dM, *dT, *dR, *dTR;
    uint8_t *dflip;
    double *dsum;
    int32_t *dflag;
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    HIP_OK(hipMalloc((void **)&dM, bytes)); HIP_OK(hipMalloc((void **)&dT, bytes));
    HIP_OK(hipMalloc((void **)&dR, bytes)); HIP_OK(hipMalloc((void **)&dTR, bytes));
    HIP_OK(hipMalloc((void **)&dflip, (size_t)B)); HIP_OK(hipMalloc((void **)&dsum, 2 * sizeof(double)));
    HIP_OK(hipMalloc((void **)&dflag, sizeof(int32_t)));
    HIP_OK(hipMemcpyAsync(dM, hM, bytes, hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(dT, hT, bytes, hipMemcpyHostToDevice, stream));

    if (so3_version() != SO3PROJ_VERSION) {                         /* argument lists changed between versions without new names for every symbol */
        fprintf(stderr, "libso3proj.so is version %d, this program was compiled against %d\n", so3_version(), SO3PROJ_VERSION);
        return 4;
    }
    printf("libso3proj version %d, %lld rows\n", so3_version(), (long long)B);
    SO3_OK(so3_project_fwd_f32(dM, dR, dflip, B, stream));          /* R = proj(M), flip = det(M) < 0 */
    SO3_OK(so3_project_fwd_f32(dT, dTR, NULL, B, stream));          /* a second batch of rotations */
    SO3_OK(so3_angle_error_v2(dR, dTR, NULL, dsum, dflag, NULL, 0u, B, stream));   /* degrees; no workspace: the call zeroes (sum, count) itself */

    double sum_count[2];
    int32_t flag;
    HIP_OK(hipMemcpyAsync(hR, dR, bytes, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(hflip, dflip, (size_t)B, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(sum_count, dsum, sizeof sum_count, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(&flag, dflag, sizeof flag, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));

    double worst_orth = 0.0, worst_det = 0.0;
    int64_t flips = 0, flip_mismatch = 0;
    for (int64_t b = 0; b < B; ++b) {
        const float *r = hR + 9 * b, *m = hM + 9 * b;
        double e = 0.0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double d = (i == j) ? -1.0 : 0.0;
                for (int k = 0; k < 3; ++k) d += (double)r[3 * k + i] * r[3 * k + j];
                e += d * d;
            }
        if (sqrt(e) > worst_orth) worst_orth = sqrt(e);
        const double det = (double)r[0] * ((double)r[4] * r[8] - (double)r[5] * r[7]) - (double)r[1] * ((double)r[3] * r[8] - (double)r[5] * r[6])
                           + (double)r[2] * ((double)r[3] * r[7] - (double)r[4] * r[6]);
        if (fabs(det - 1.0) > worst_det) worst_det = fabs(det - 1.0);
        const double dm = (double)m[0] * ((double)m[4] * m[8] - (double)m[5] * m[7]) - (double)m[1] * ((double)m[3] * m[8] - (double)m[5] * m[6])
                          + (double)m[2] * ((double)m[3] * m[7] - (double)m[4] * m[6]);
        flips += hflip[b];
        flip_mismatch += (hflip[b] != (dm < 0.0));
    }
    const double mean_angle = sum_count[0] / sum_count[1];
    printf("max ||R^T R - I||_F = %.3e, max |det R - 1| = %.3e, flips = %lld (mismatch vs det(M) < 0: %lld)\n", worst_orth, worst_det,
           (long long)flips, (long long)flip_mismatch);
    printf("mean geodesic angle between two random projections = %.4f deg (Haar expectation 126.476), count = %.0f, range flag = %d\n",
           mean_angle, sum_count[1], flag);
    const int ok = worst_orth < 1e-5 && worst_det < 1e-5 && flip_mismatch == 0 && flag == 0 && sum_count[1] == (double)B
                   && (B < 50000 || fabs(mean_angle - 126.476) < 1.0);      /* the Haar mean needs a large sample */
    /* invalid arguments come back as codes, never as exceptions */
    const int rc = so3_project_fwd_f32(NULL, dR, NULL, B, stream);
    printf("so3_project_fwd_f32(NULL, ...) -> %d (%s)\n", rc, so3_last_error());
    puts(ok && rc != 0 ? "OK" : "FAILED");
    return ok && rc != 0 ? 0 : 1;
}
