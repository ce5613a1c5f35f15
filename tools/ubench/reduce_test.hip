// Unit check of wave_reduce32 (photo_icp_kernels.h): every lane must be counted exactly once for every value.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../rgbd360_amd/csrc/photo_icp_kernels.h"
using namespace r360;
__global__ void k(float* res /*[64 L0][32 k]*/) {
    const int lane = threadIdx.x;
    for (int L0 = 0; L0 < 64; ++L0) {
        float v[32], out[2];
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) v[kk] = (lane == L0) ? (float)(kk + 1) : 0.f;
        wave_reduce32(v, out);
        if ((lane & 3) == 0) {
            const int row = lane >> 4, quad = (lane >> 2) & 3;
            const int idx = 2 * (quad & 1) + 4 * (quad >> 1) + 8 * (row & 1) + 16 * (row >> 1);
            res[L0 * 32 + idx] = out[0];
            res[L0 * 32 + idx + 1] = out[1];
        }
        __syncthreads();
    }
}
int main() {
    float* d; hipMalloc(&d, 64 * 32 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    std::vector<float> h(64 * 32);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int L0 = 0; L0 < 64; ++L0) {
        printf("L0=%2d ratios:", L0);
        for (int kk = 0; kk < 32; ++kk) {
            printf(" %g", h[L0 * 32 + kk] / (kk + 1));
            if (h[L0 * 32 + kk] != (float)(kk + 1)) ++bad;
        }
        printf("\n");
    }
    printf("bad %d of 2048\n", bad);
    return bad != 0;
}
