// Micro-benchmark 3: v_cndmask_b32 reading VCC (VOP2) versus an SGPR pair (VOP3), by WHO wrote the mask last.
// Build: hipcc --offload-arch=gfx950 -O3 -o vcc_select vcc_select.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float y0 = 0, y1 = 0, y2 = 0, y3 = 0, y4 = 0, y5 = 0, y6 = 0, y7 = 0;
    unsigned long long m = __ballot(threadIdx.x & 1), m2 = __ballot(threadIdx.x & 2);
    if (KIND == 0 || KIND == 5) asm volatile("s_mov_b64 vcc, %0" :: "s"(m) : "vcc", "scc");
    if (KIND == 1) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(x0), "v"(a) : "vcc", "scc");
#define SEL8_VCC "v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %9, %10, vcc\n v_cndmask_b32 %2, %10, %11, vcc\n v_cndmask_b32 %3, %11, %12, vcc\n" \
                 "v_cndmask_b32 %4, %12, %13, vcc\n v_cndmask_b32 %5, %13, %14, vcc\n v_cndmask_b32 %6, %14, %15, vcc\n v_cndmask_b32 %7, %15, %8, vcc\n"
#define SEL8_SG(r) "v_cndmask_b32 %0, %8, %9, " r "\n v_cndmask_b32 %1, %9, %10, " r "\n v_cndmask_b32 %2, %10, %11, " r "\n v_cndmask_b32 %3, %11, %12, " r "\n" \
                   "v_cndmask_b32 %4, %12, %13, " r "\n v_cndmask_b32 %5, %13, %14, " r "\n v_cndmask_b32 %6, %14, %15, " r "\n v_cndmask_b32 %7, %15, %8, " r "\n"
#define OUTS : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7)
#define INS  "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7)
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {          // vcc written by s_mov before the loop
            asm volatile(SEL8_VCC OUTS : INS);
        } else if (KIND == 1) {   // vcc written by v_cmp before the loop
            asm volatile(SEL8_VCC OUTS : INS);
        } else if (KIND == 2) {   // s_and_b64 vcc in the loop, then 8 selects
            asm volatile("s_and_b64 vcc, %16, %17\n" SEL8_VCC OUTS : INS, "s"(m), "s"(m2) : "vcc", "scc");
        } else if (KIND == 3) {   // s_and_b64 into an SGPR pair in the loop, then 8 VOP3 selects
            asm volatile("s_and_b64 s[20:21], %16, %17\n s_nop 0\n" SEL8_SG("s[20:21]") OUTS : INS, "s"(m), "s"(m2) : "s20", "s21", "scc");
        } else if (KIND == 4) {   // v_cmp vcc in the loop, then 8 selects
            asm volatile("v_cmp_gt_f32 vcc, %8, %16\n" SEL8_VCC OUTS : INS, "v"(a) : "vcc", "scc");
        } else if (KIND == 5) {   // vcc from s_mov, selects written as VOP3 (vcc named as an SGPR-pair operand, e64)
            asm volatile("v_cndmask_b32_e64 %0, %8, %9, vcc\n v_cndmask_b32_e64 %1, %9, %10, vcc\n v_cndmask_b32_e64 %2, %10, %11, vcc\n v_cndmask_b32_e64 %3, %11, %12, vcc\n"
                         "v_cndmask_b32_e64 %4, %12, %13, vcc\n v_cndmask_b32_e64 %5, %13, %14, vcc\n v_cndmask_b32_e64 %6, %14, %15, vcc\n v_cndmask_b32_e64 %7, %15, %8, vcc\n" OUTS : INS);
        } else if (KIND == 6) {   // v_cmp into an SGPR pair, s_and with another pair into vcc, 8 selects (the compiler's idiom)
            asm volatile("v_cmp_gt_f32 s[20:21], %8, %17\n s_and_b64 vcc, s[20:21], %16\n" SEL8_VCC OUTS : INS, "s"(m), "v"(a) : "vcc", "s20", "s21", "scc");
        } else if (KIND == 7) {   // same idiom kept in SGPR pairs (VOP3 selects)
            asm volatile("v_cmp_gt_f32 s[20:21], %8, %17\n s_and_b64 s[20:21], s[20:21], %16\n s_nop 0\n" SEL8_SG("s[20:21]") OUTS : INS, "s"(m), "v"(a) : "s20", "s21", "scc");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7;
}

template <int KIND>
void run(const char* name, int blocks_per_cu) {
    float* out;
    int blocks = 256 * blocks_per_cu;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 1e-6f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 1e-6f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winst = (double)iters * 8 * blocks_per_cu;
    printf("%-64s waves/SIMD=%d  %7.3f ms -> %.2f ns / select / SIMD\n", name, blocks_per_cu, ms, ms * 1e6 / winst);
    hipFree(out);
}

int main() {
    for (int w : {1, 4}) {
        run<0>("VOP2 select, vcc from s_mov before the loop", w);
        run<1>("VOP2 select, vcc from v_cmp before the loop", w);
        run<5>("VOP3 select naming vcc, vcc from s_mov before the loop", w);
        run<2>("s_and_b64 vcc + 8 VOP2 selects per trip", w);
        run<3>("s_and_b64 s[n:n+1] + 8 VOP3 selects per trip", w);
        run<4>("v_cmp vcc + 8 VOP2 selects per trip", w);
        run<6>("v_cmp s[], s_and_b64 vcc + 8 VOP2 selects per trip", w);
        run<7>("v_cmp s[], s_and_b64 s[] + 8 VOP3 selects per trip", w);
    }
    return 0;
}
