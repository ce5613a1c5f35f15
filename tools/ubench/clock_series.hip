// Micro-benchmark (round 6): the shader clock over a burst of back-to-back full-chip launches, after an idle gap.  Every launch runs the same
// fixed amount of work -- a stream over a 256 MB buffer with 64 fused multiply-adds per element, HBM and vector pipes both busy, like the
// batch pass k_eval_b -- and reports its own duration (HIP events) and the shader clock it ran at (shader cycles / 100 MHz real-time ticks,
// from one wave per block).  Answers VERDICT r5 item 6: is the 151 -> 204 us spread of k_eval_b's launches the clock?
// Build: hipcc --offload-arch=gfx950 -O3 -o clock_series clock_series.hip ; run: ./clock_series [launches per burst] [idle ms]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

template <int K>
__global__ __launch_bounds__(1024) void k_work(const float4* __restrict__ in, float4* __restrict__ out, size_t n, unsigned long long* stamps) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = in[i];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            v.x = fmaf(v.x, 1.0000001f, 1e-7f); v.y = fmaf(v.y, 0.9999999f, 1e-7f);
            v.z = fmaf(v.z, 1.0000001f, -1e-7f); v.w = fmaf(v.w, 0.9999999f, -1e-7f);
        }
        out[i] = v;
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = c1 - c0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

int main(int argc, char** argv) {
    const int n_launch = argc > 1 ? atoi(argv[1]) : 24;
    const int idle_ms = argc > 2 ? atoi(argv[2]) : 10;
    const size_t n = (size_t)16 << 20;                      // 16 M float4 = 256 MB in, 256 MB out: past the Infinity Cache
    float4 *in, *out;
    unsigned long long* st;
    const int blocks = 512;
    hipMalloc(&in, n * sizeof(float4));
    hipMalloc(&out, n * sizeof(float4));
    hipMalloc(&st, (size_t)n_launch * blocks * 2 * sizeof(unsigned long long));
    hipMemset(in, 0, n * sizeof(float4));
    std::vector<hipEvent_t> ev(n_launch + 1);
    for (auto& e : ev) hipEventCreate(&e);
    hipStream_t s;
    hipStreamCreate(&s);
    for (int burst = 0; burst < 4; ++burst) {
        const bool heavy = burst >= 2;          // bursts 0-1: 64 fused multiply-adds per 32 bytes (bound by HBM); 2-3: 448 (bound by vector issue, HBM still streaming)
        hipStreamSynchronize(s);
        std::this_thread::sleep_for(std::chrono::milliseconds(idle_ms));
        hipEventRecord(ev[0], s);
        for (int l = 0; l < n_launch; ++l) {
            if (heavy) hipLaunchKernelGGL(k_work<112>, dim3(blocks), dim3(1024), 0, s, in, out, n, st + (size_t)l * blocks * 2);
            else hipLaunchKernelGGL(k_work<16>, dim3(blocks), dim3(1024), 0, s, in, out, n, st + (size_t)l * blocks * 2);
            hipEventRecord(ev[l + 1], s);
        }
        hipStreamSynchronize(s);
        std::vector<unsigned long long> h((size_t)n_launch * blocks * 2);
        hipMemcpy(h.data(), st, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        printf("burst %d (%s, after %d ms idle): launch us | shader MHz\n", burst, heavy ? "issue-bound" : "HBM-bound", idle_ms);
        for (int l = 0; l < n_launch; ++l) {
            float ms = 0;
            hipEventElapsedTime(&ms, ev[l], ev[l + 1]);
            double cyc = 0, ticks = 0;
            for (int b = 0; b < blocks; ++b) { cyc += (double)h[((size_t)l * blocks + b) * 2]; ticks += (double)h[((size_t)l * blocks + b) * 2 + 1]; }
            printf("  %2d: %7.1f us | %5.0f MHz\n", l, ms * 1e3, cyc / ticks * 100.0);
        }
    }
    return 0;
}
