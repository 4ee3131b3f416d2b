// occupancy: how many workgroups of the engine's kernels the runtime says fit one CU (registers, LDS), next to what they are launched for.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -o occupancy occupancy.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "../../poseestimation_amd/csrc/so3_rows.h"

template <class Op, int NPL, int WPS> void report(const char *name) {
    int blocks = 0;
    hipFuncAttributes a;
    auto *fn = &so3::k_rows<Op, NPL, WPS, 256, false>;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fn, 256, 0);
    hipFuncGetAttributes(&a, reinterpret_cast<const void *>(fn));
    printf("%-44s built for %d waves/SIMD: %d workgroups of 256 per CU = %d waves/SIMD   (VGPR %d, LDS %zu B, scratch %zu B)\n", name, WPS, blocks, blocks,
           a.numRegs, a.sharedSizeBytes, a.localSizeBytes);
}

int main() {
    report<so3::OpProject<4, false>, 2, 3>("K1 OpProject<4,false>");
    report<so3::OpProjectBwd<4>, 2, 2>("K2 OpProjectBwd<4>");
    report<so3::OpProjectBwd<4>, 2, 3>("K2 OpProjectBwd<4>");
    report<so3::OpFrobHead<4, true, true>, 2, 2>("K3 OpFrobHead<4,dM,R>");
    report<so3::OpFrobHead<4, true, true>, 2, 3>("K3 OpFrobHead<4,dM,R>");
    report<so3::OpProjectAngle<4, false, false, true, true>, 2, 2>("K1+K4 OpProjectAngle<4,sum,f32>");
    report<so3::OpProjectAngle<4, false, false, true, true>, 2, 3>("K1+K4 OpProjectAngle<4,sum,f32>");
    for (size_t extra : {size_t(0), size_t(6144), size_t(8276), size_t(10244), size_t(11600)}) {        // does the LDS a two-input kernel would need at three waves fit three times?
        int blocks = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, &so3::k_rows<so3::OpProject<4, false>, 2, 3, 256, false>, 256, extra);
        printf("K1 (168 VGPRs) with %zu B of LDS per workgroup: %d workgroups per CU\n", size_t(43012) + extra, blocks);
    }
    int lds = 0, cu = 0;
    hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, 0);
    hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, 0);
    printf("device: %d CUs, %d B of LDS per CU\n", cu, lds);
    return 0;
}
