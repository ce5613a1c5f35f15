// Does a `buffer_load_dwordx4 ... lds` (LDS-DMA, 1 KiB per wave-instruction) land correctly at LDS byte addresses beyond 64 KiB, and does a
// descriptor of zero records leave its LDS piece untouched?  (The batched form of the per-pixel pass parks a 1024-thread block's source
// records -- 9 steps x 16 KiB = 144 KiB -- in LDS while the solve prologue runs.)
//   hipcc --offload-arch=gfx950 -O3 -o lds_dma_records lds_dma_records.hip && ./lds_dma_records
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int kSteps = 9;
__global__ __launch_bounds__(1024) void k(const float4* __restrict__ src, float4* __restrict__ out, int n, int null_step) {
    __shared__ float4 buf[kSteps][1024];
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < kSteps; ++s) buf[s][tid] = make_float4(-1.f, -2.f, -3.f, -4.f);
    __syncthreads();
    const float4* base = src + (size_t)blockIdx.x * kSteps * 1024;
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(base + s * 1024), 0, s == null_step ? 0 : 1024 * 16, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)&buf[s][tid & ~63], 16, (tid & 1023) * 16, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < kSteps; ++s) out[((size_t)blockIdx.x * kSteps + s) * 1024 + tid] = buf[s][tid];
}
int main() {
    const int blocks = 256, n = blocks * kSteps * 1024;
    std::vector<float4> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = make_float4((float)i, (float)(i ^ 0x55), (float)(i >> 3), (float)(i * 3));
    float4 *d, *dout;
    hipMalloc(&d, n * sizeof(float4)); hipMalloc(&dout, n * sizeof(float4));
    hipMemcpy(d, h.data(), n * sizeof(float4), hipMemcpyHostToDevice);
    for (int null_step : {-1, 4, 8}) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 0, 0, d, dout, n, null_step);
        hipMemcpy(o.data(), dout, n * sizeof(float4), hipMemcpyDeviceToHost);
        long bad = 0, zero = 0, kept = 0;
        for (int i = 0; i < n; ++i) {
            const int s = (i / 1024) % kSteps;
            const float4 want = h[i];
            if (s == null_step) {
                if (o[i].x == -1.f && o[i].w == -4.f) ++kept;
                else if (o[i].x == 0.f && o[i].y == 0.f && o[i].z == 0.f && o[i].w == 0.f) ++zero;
                else ++bad;
            } else if (o[i].x != want.x || o[i].y != want.y || o[i].z != want.z || o[i].w != want.w) ++bad;
        }
        printf("null_step %d: mismatches %ld; null piece: kept-old %ld, zero-filled %ld (err %s)\n", null_step, bad, kept, zero, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
