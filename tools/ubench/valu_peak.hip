// Micro-benchmark: what one SIMD of gfx950 really issues per cycle, measured in SHADER cycles (s_memtime) next to wall time
// (s_memrealtime, 100 MHz), for 1 / 2 / 4 / 8 waves per SIMD and independent instruction streams.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_peak valu_peak.hip ; run: ./valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* stamps, int iters, float a, float b) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
    double d0 = x0, d1 = x1, d2 = x2, d3 = x3;
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (KIND == 1) {
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
        } else if (KIND == 2) {
            asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)a));
        } else if (KIND == 3) {
            asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                         "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : : );
        } else if (KIND == 4) {   // fma with a scalar operand (VOP3 + SGPR)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a), "v"(b));
        } else if (KIND == 5) {   // VOP2 mul / fmac
            asm volatile("v_mul_f32 %0, %0, %8\n v_fmac_f32 %1, %0, %8\n v_mul_f32 %2, %2, %8\n v_fmac_f32 %3, %2, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_fmac_f32 %5, %4, %8\n v_mul_f32 %6, %6, %8\n v_fmac_f32 %7, %6, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (KIND == 6) {   // v_fma_f64
            asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)a));
        }
    }
    unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(d0 + d1 + d2 + d3);
    if ((threadIdx.x & 63) == 0) {
        int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = c1 - c0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

template <int KIND>
void run(const char* name, int per_iter, int blocks_per_cu, int n_cu) {
    float* out;
    unsigned long long* st;
    int blocks = n_cu * blocks_per_cu;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&st, blocks * 4 * 2 * sizeof(unsigned long long));
    int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, st, iters, 1.0001f, 1e-6f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, st, iters, 1.0001f, 1e-6f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc, rt;
    for (int w = 0; w < blocks * 4; ++w) { cyc.push_back((double)h[2 * w]); rt.push_back((double)h[2 * w + 1]); }
    std::sort(cyc.begin(), cyc.end()); std::sort(rt.begin(), rt.end());
    double cmed = cyc[cyc.size() / 2], rmed = rt[rt.size() / 2];     // s_memtime ticks, 100 MHz ticks
    double winst = (double)iters * per_iter * blocks_per_cu;          // wave-instructions per SIMD
    printf("%-14s CUs=%3d waves/SIMD=%d  %.3f ms | per wave: %.0f memtime ticks in %.1f us = %.0f MHz | %.2f memtime ticks / wave-instr / SIMD | %.2f ns / wave-instr / SIMD\n",
           name, n_cu, blocks_per_cu, ms, cmed, rmed * 0.01, cmed / (rmed * 0.01), cmed / winst, rmed * 10.0 / winst);
    hipFree(out); hipFree(st);
}

int main() {
    for (int n_cu : {256, 32}) {
        for (int w : {1, 2, 4, 8}) {
            run<0>("v_fma_f32", 8, w, n_cu);
            run<1>("v_pk_fma_f32", 4, w, n_cu);
            run<4>("v_fma_f32 sgpr", 8, w, n_cu);
            run<5>("v_mul/fmac", 8, w, n_cu);
            run<3>("v_cndmask", 8, w, n_cu);
            run<2>("v_add_f64", 4, w, n_cu);
            run<6>("v_fma_f64", 4, w, n_cu);
        }
    }
    return 0;
}
