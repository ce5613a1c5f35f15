// What would the pass's blocks pay for adding their partial rows into 32 global fixed-point accumulators at their end (integer atomics:
// order-free, reproducible) instead of writing a row each -- and what would the next launch's prologue save by loading 32 totals instead
// of 256 rows?  256 blocks x 1024 threads; each block spins ~10 us (a stand-in for the pass), then either stores its row (32 doubles)
// or issues 32 64-bit atomic adds (accumulators one cache line apart); the next launch's prologue either loads + sums 256 rows the way
// stage_pending does or loads the 32 totals.  Back-to-back launches, time per launch from HIP events.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/tail_atomics tools/ubench/tail_atomics.hip && /tmp/tail_atomics
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr int kThreads = 1024, kVals = 32, kRows = 256, kPad = 32;      // accumulators 256 B apart

template <int MODE>      // 0: rows out, rows in; 1: atomics out, totals in; 2: nothing in, nothing out (the spin alone)
__global__ __launch_bounds__(kThreads) void k_iter(const double* __restrict__ rows_in, double* __restrict__ rows_out,
                                                   const unsigned long long* __restrict__ acc_in, unsigned long long* __restrict__ acc_out,
                                                   unsigned long long* __restrict__ acc_clear, int spin_ticks, double* __restrict__ sink) {
    __shared__ double red[kThreads / 64][kVals];
    __shared__ double tot[kVals];
    const int tid = threadIdx.x, v = tid % kVals, q = tid / kVals;
    double s = 0.0;
    if (MODE == 0) {
        double tmp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) tmp[j] = rows_in[(size_t)(q + 32 * j) * kVals + v];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += tmp[j];
        const double other = __shfl_xor(s, 32);
        if ((tid & 63) < kVals) red[tid >> 6][tid & 31] = s + other;
        __syncthreads();
        if (tid < kVals) {
            double t = 0.0;
            for (int k = 0; k < kThreads / 64; ++k) t += red[k][tid];
            tot[tid] = t;
        }
    } else if (MODE == 1) {
        if (tid < kVals) tot[tid] = (double)(long long)acc_in[tid * kPad] * (1.0 / 1024.0);
        if (blockIdx.x == 0 && tid < kVals) acc_clear[tid * kPad] = 0ull;      // the buffer of the launch after the next
    } else {
        if (tid < kVals) tot[tid] = 1.0;
    }
    __syncthreads();
    const double pose = tot[0] + tot[5] + tot[31];
    // the "pass"
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(4);
    if (tid < kVals) {
        const double val = pose * 1e-9 + (double)(blockIdx.x + tid);
        if (MODE == 0) rows_out[(size_t)blockIdx.x * kVals + tid] = val;
        else if (MODE == 1) atomicAdd(&acc_out[tid * kPad], (unsigned long long)(long long)(val * 1024.0));
    }
    if (blockIdx.x == 0 && tid == 0) *sink = pose;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    double *rows[2], *sink; unsigned long long* acc[3];
    for (int k = 0; k < 2; ++k) { CK(hipMalloc(&rows[k], kRows * kVals * 8)); CK(hipMemset(rows[k], 0, kRows * kVals * 8)); }
    for (int k = 0; k < 3; ++k) { CK(hipMalloc(&acc[k], kVals * kPad * 8)); CK(hipMemset(acc[k], 0, kVals * kPad * 8)); }
    CK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int spin_us : {10, 0}) {
        float ms[3] = {0, 0, 0};
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                for (int it = 0; it < iters; ++it) {
                    const int spin = spin_us * 100;
                    if (mode == 0) hipLaunchKernelGGL(k_iter<0>, dim3(kRows), dim3(kThreads), 0, 0, rows[it & 1], rows[(it + 1) & 1], acc[0], acc[1], acc[2], spin, sink);
                    else if (mode == 1) hipLaunchKernelGGL(k_iter<1>, dim3(kRows), dim3(kThreads), 0, 0, rows[0], rows[1], acc[it % 3], acc[(it + 1) % 3], acc[(it + 2) % 3], spin, sink);
                    else hipLaunchKernelGGL(k_iter<2>, dim3(kRows), dim3(kThreads), 0, 0, rows[0], rows[1], acc[0], acc[1], acc[2], spin, sink);
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[mode], e0, e1));
            }
        }
        printf("pass stand-in %2d us: rows out + 256 rows in %6.2f us / launch | 32 atomics out + 32 totals in %6.2f | neither %6.2f\n", spin_us,
               ms[0] * 1e3 / iters, ms[1] * 1e3 / iters, ms[2] * 1e3 / iters);
    }
    return 0;
}
