// Exhaustive check of cheaper correctly-rounded sqrt / reciprocal candidates against the compiler's IEEE sqrtf and 1.f/x over
// the bit range the warp front end uses ([2^-60, 2^60]).  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o rn_variants
// rn_variants.hip ; run: ./rn_variants
#include <hip/hip_runtime.h>
#include <cstdio>

template <int V>
__device__ __forceinline__ float sqrt_v(float x) {
    if (V == 0) {   // current: hardware sqrt + two exact residual tests
        float s = __builtin_amdgcn_sqrtf(x);
        const float sm = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1);
        const float sp = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
        const float rm = fmaf(-sm, s, x), rp = fmaf(-sp, s, x);
        s = (rm <= 0.f) ? sm : s;
        s = (rp > 0.f) ? sp : s;
        return s;
    } else if (V == 1) {   // rsq + one coupled Newton step
        const float y = __builtin_amdgcn_rsqf(x);
        float s = x * y;
        const float h = 0.5f * y;
        const float r = fmaf(-s, s, x);
        return fmaf(r, h, s);
    } else if (V == 2) {   // rsq + two steps
        const float y = __builtin_amdgcn_rsqf(x);
        float s = x * y;
        const float h = 0.5f * y;
        float r = fmaf(-s, s, x);
        s = fmaf(r, h, s);
        r = fmaf(-s, s, x);
        return fmaf(r, h, s);
    } else if (V == 4) {   // harness check: uncorrected estimates MUST mismatch
        return x * __builtin_amdgcn_rsqf(x);
    } else {               // hardware sqrt + one Newton step with h = 0.5 rsq
        float s = __builtin_amdgcn_sqrtf(x);
        const float h = 0.5f * __builtin_amdgcn_rsqf(x);
        const float r = fmaf(-s, s, x);
        return fmaf(r, h, s);
    }
}
template <int V>
__device__ __forceinline__ float rcp_v(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    if (V == 4) return r;
    if (V == 0) {
        float e = fmaf(-x, r, 1.f);
        r = fmaf(e, r, r);
        e = fmaf(-x, r, 1.f);
        return fmaf(e, r, r);
    } else {
        const float e = fmaf(-x, r, 1.f);
        return fmaf(e, r, r);
    }
}

template <int V>
__global__ void k(unsigned first, unsigned count, unsigned long long* bad) {
    unsigned long long bs = 0, br = 0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const float x = __builtin_bit_cast(float, first + i);
        if (sqrt_v<V>(x) != sqrtf(x)) ++bs;
        if (rcp_v<V>(x) != 1.f / x) ++br;
        if (rcp_v<V>(-x) != 1.f / -x) ++br;
    }
    if (bs) atomicAdd(&bad[0], bs);
    if (br) atomicAdd(&bad[1], br);
}

int main() {
    unsigned long long* bad;
    hipMalloc(&bad, 16);
    const unsigned first = 0x21800000u, last = 0x5D800000u;
    for (int v = 0; v < 5; ++v) {
        hipMemset(bad, 0, 16);
        if (v == 0) hipLaunchKernelGGL(k<0>, dim3(4096), dim3(256), 0, 0, first, last - first, bad);
        if (v == 1) hipLaunchKernelGGL(k<1>, dim3(4096), dim3(256), 0, 0, first, last - first, bad);
        if (v == 2) hipLaunchKernelGGL(k<2>, dim3(4096), dim3(256), 0, 0, first, last - first, bad);
        if (v == 4) hipLaunchKernelGGL(k<4>, dim3(4096), dim3(256), 0, 0, first, last - first, bad);
        if (v == 3) hipLaunchKernelGGL(k<3>, dim3(4096), dim3(256), 0, 0, first, last - first, bad);
        unsigned long long h[2];
        if (hipDeviceSynchronize() != hipSuccess) printf("kernel failed\n");
        hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
        printf("variant %d: sqrt mismatches %llu, rcp mismatches %llu (of %u values)\n", v, h[0], h[1], last - first);
    }
    return 0;
}
