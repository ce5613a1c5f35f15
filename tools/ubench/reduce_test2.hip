#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../rgbd360_amd/csrc/photo_icp_kernels.h"
using namespace r360;
__global__ void k(float* o, int L0) {
    const int lane = threadIdx.x;
    float v[32];
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) v[kk] = (lane == L0) ? (float)(kk + 1) : 0.f;
    float u[16], t[8];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v[j]), __builtin_bit_cast(unsigned, v[j + 16]), false, false);
        u[j] = __builtin_bit_cast(float, sw[0]) + __builtin_bit_cast(float, sw[1]);
    }
    o[lane] = u[0]; o[64 + lane] = u[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, u[j]), __builtin_bit_cast(unsigned, u[j + 8]), false, false);
        t[j] = __builtin_bit_cast(float, sw[0]) + __builtin_bit_cast(float, sw[1]);
        if (j == 0) { o[128 + lane] = __builtin_bit_cast(float, sw[0]); o[192 + lane] = __builtin_bit_cast(float, sw[1]); }
    }
    o[256 + lane] = t[0];
}
int main() {
    float* d; hipMalloc(&d, 320 * 4);
    for (int L0 : {0, 16}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, L0);
        float h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        const char* names[5] = {"u[0]", "u[8]", "sw16[0]", "sw16[1]", "t[0]"};
        printf("L0=%d\n", L0);
        for (int t = 0; t < 5; ++t) { printf("%-8s", names[t]); for (int l = 0; l < 64; ++l) printf(" %g", h[t * 64 + l]); printf("\n"); }
    }
    return 0;
}
