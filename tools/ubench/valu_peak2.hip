// Micro-benchmark 2: per-SIMD issue cost of instruction FORMS (operand kinds), whole chip busy, 4 and 8 waves per SIMD.
// cost = launch time (HIP events) / wave-instructions per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_peak2 valu_peak2.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(s) s s s s s s s s

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float y0 = 0, y1 = 0, y2 = 0, y3 = 0, y4 = 0, y5 = 0, y6 = 0, y7 = 0;
    unsigned long long m = __ballot(threadIdx.x & 1);
    asm volatile("s_mov_b64 vcc, %0" :: "s"(m) : "vcc");
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {         // reference: v_fma_f32, all VGPR
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (KIND == 1) {  // cndmask, 3-address, vcc (VOP2)
            asm volatile("v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %9, %10, vcc\n v_cndmask_b32 %2, %10, %11, vcc\n v_cndmask_b32 %3, %11, %12, vcc\n"
                         "v_cndmask_b32 %4, %12, %13, vcc\n v_cndmask_b32 %5, %13, %14, vcc\n v_cndmask_b32 %6, %14, %15, vcc\n v_cndmask_b32 %7, %15, %8, vcc\n"
                         : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7)
                         : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
        } else if (KIND == 2) {  // cndmask, dst = src0, vcc (the form of valu_peak.hip)
            asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                         "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : : );
        } else if (KIND == 3) {  // cndmask with an SGPR-pair mask (VOP3)
            asm volatile("v_cndmask_b32 %0, %8, %9, %16\n v_cndmask_b32 %1, %9, %10, %16\n v_cndmask_b32 %2, %10, %11, %16\n v_cndmask_b32 %3, %11, %12, %16\n"
                         "v_cndmask_b32 %4, %12, %13, %16\n v_cndmask_b32 %5, %13, %14, %16\n v_cndmask_b32 %6, %14, %15, %16\n v_cndmask_b32 %7, %15, %8, %16\n"
                         : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7)
                         : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7), "s"(m));
        } else if (KIND == 4) {  // VOP2 mul with an SGPR operand
            asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                         "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a));
        } else if (KIND == 5) {  // VOP2 mul, all VGPR
            asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                         "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (KIND == 6) {  // VOP2 mul with a 32-bit literal
            asm volatile("v_mul_f32 %0, 0x3f800347, %0\n v_mul_f32 %1, 0x3f800347, %1\n v_mul_f32 %2, 0x3f800347, %2\n v_mul_f32 %3, 0x3f800347, %3\n"
                         "v_mul_f32 %4, 0x3f800347, %4\n v_mul_f32 %5, 0x3f800347, %5\n v_mul_f32 %6, 0x3f800347, %6\n v_mul_f32 %7, 0x3f800347, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (KIND == 7) {  // v_fmaak (literal addend)
            asm volatile("v_fmaak_f32 %0, %8, %0, 0x358637bd\n v_fmaak_f32 %1, %8, %1, 0x358637bd\n v_fmaak_f32 %2, %8, %2, 0x358637bd\n v_fmaak_f32 %3, %8, %3, 0x358637bd\n"
                         "v_fmaak_f32 %4, %8, %4, 0x358637bd\n v_fmaak_f32 %5, %8, %5, 0x358637bd\n v_fmaak_f32 %6, %8, %6, 0x358637bd\n v_fmaak_f32 %7, %8, %7, 0x358637bd\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (KIND == 8) {  // v_cmp to vcc
            asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n"
                         "v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");
        } else if (KIND == 9) {  // v_cmp to an SGPR pair (VOP3)
            unsigned long long t;
            asm volatile("v_cmp_gt_f32 %8, %0, %9\n v_cmp_gt_f32 %8, %1, %9\n v_cmp_gt_f32 %8, %2, %9\n v_cmp_gt_f32 %8, %3, %9\n"
                         "v_cmp_gt_f32 %8, %4, %9\n v_cmp_gt_f32 %8, %5, %9\n v_cmp_gt_f32 %8, %6, %9\n v_cmp_gt_f32 %8, %7, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "=&s"(t) : "v"(a));
        } else if (KIND == 10) { // v_mov_b32
            asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %9\n v_mov_b32 %2, %10\n v_mov_b32 %3, %11\n v_mov_b32 %4, %12\n v_mov_b32 %5, %13\n v_mov_b32 %6, %14\n v_mov_b32 %7, %15\n"
                         : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7)
                         : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
        } else if (KIND == 11) { // v_readlane to SGPR (lane select constant)
            int s;
            asm volatile("v_readlane_b32 %8, %0, 3\n v_readlane_b32 %8, %1, 3\n v_readlane_b32 %8, %2, 3\n v_readlane_b32 %8, %3, 3\n"
                         "v_readlane_b32 %8, %4, 3\n v_readlane_b32 %8, %5, 3\n v_readlane_b32 %8, %6, 3\n v_readlane_b32 %8, %7, 3\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "=&s"(s));
        } else if (KIND == 12) { // cmp + cndmask pairs (cmp to vcc, cndmask reads it): the select idiom
            asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %0, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %2, vcc\n"
                         "v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %4, vcc\n v_cmp_gt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %6, vcc\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");
        } else if (KIND == 13) { // v_max_f32 / v_min_f32 (select-free alternative)
            asm volatile("v_max_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n"
                         "v_max_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (KIND == 14) { // v_max_u32
            asm volatile("v_max_u32 %0, %0, %8\n v_max_u32 %1, %1, %8\n v_max_u32 %2, %2, %8\n v_max_u32 %3, %3, %8\n"
                         "v_max_u32 %4, %4, %8\n v_max_u32 %5, %5, %8\n v_max_u32 %6, %6, %8\n v_max_u32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (KIND == 15) { // cndmask with inline constants as sources (0 / 1.0)
            asm volatile("v_cndmask_b32 %0, 0, %8, vcc\n v_cndmask_b32 %1, 0, %9, vcc\n v_cndmask_b32 %2, 0, %10, vcc\n v_cndmask_b32 %3, 0, %11, vcc\n"
                         "v_cndmask_b32 %4, 0, %12, vcc\n v_cndmask_b32 %5, 0, %13, vcc\n v_cndmask_b32 %6, 0, %14, vcc\n v_cndmask_b32 %7, 0, %15, vcc\n"
                         : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7)
                         : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
        } else if (KIND == 16) { // fma with vcc all-ones EXEC... plain v_add_f32
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (KIND == 17) { // integer: v_add_u32 / v_lshlrev / v_and
            asm volatile("v_add_u32 %0, %0, %8\n v_lshlrev_b32 %1, 1, %1\n v_and_b32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                         "v_lshlrev_b32 %4, 1, %4\n v_and_b32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7;
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
    printf("%-34s waves/SIMD=%d  %7.3f ms -> %.2f ns / wave-instr / SIMD\n", name, blocks_per_cu, ms, ms * 1e6 / winst);
    hipFree(out);
}

int main() {
    for (int w : {4, 8}) {
        run<0>("v_fma_f32 vgpr", w);
        run<16>("v_add_f32 vgpr", w);
        run<5>("v_mul_f32 vgpr (VOP2)", w);
        run<4>("v_mul_f32 sgpr (VOP2)", w);
        run<6>("v_mul_f32 literal", w);
        run<7>("v_fmaak_f32 literal", w);
        run<10>("v_mov_b32", w);
        run<17>("int add/shift/and/xor", w);
        run<13>("v_max/min_f32", w);
        run<14>("v_max_u32", w);
        run<1>("v_cndmask vcc 3-address", w);
        run<2>("v_cndmask vcc dst=src0", w);
        run<3>("v_cndmask sgpr-pair mask (VOP3)", w);
        run<15>("v_cndmask vcc, inline 0 source", w);
        run<8>("v_cmp -> vcc", w);
        run<9>("v_cmp -> sgpr pair", w);
        run<12>("v_cmp vcc + v_cndmask pairs", w);
        run<11>("v_readlane_b32", w);
    }
    return 0;
}
