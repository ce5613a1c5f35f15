#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    const unsigned lane = threadIdx.x;
    unsigned a = lane, b = 100 + lane;
    auto s16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    auto s32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[lane] = s16[0]; o[64 + lane] = s16[1]; o[128 + lane] = s32[0]; o[192 + lane] = s32[1];
    // masked DPP adds
    float x = (float)lane, d = 0.f;
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3" : "+v"(d) : "v"(x));
    o[256 + lane] = (unsigned)d;
    d = 0.f;
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5" : "+v"(d) : "v"(x));
    o[320 + lane] = (unsigned)d;
    d = 0.f;
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa" : "+v"(d) : "v"(x));
    o[384 + lane] = (unsigned)d;
}
int main() {
    unsigned* d; hipMalloc(&d, 448 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[448]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[7] = {"swap16[0]", "swap16[1]", "swap32[0]", "swap32[1]", "ror8 banks01", "shl4 banks 0,2", "shr4 banks 1,3"};
    for (int t = 0; t < 7; ++t) { printf("%-16s", names[t]); for (int l = 0; l < 64; ++l) printf(" %u", h[t * 64 + l]); printf("\n"); }
    return 0;
}
