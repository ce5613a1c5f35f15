// Exhaustive check: x / 10.0f (the compiler's IEEE division) against q = x * 0.1f, r = fma(-q, 10, x), q' = fma(r, 0.1f, q) for every
// positive float in [2^-20, 2^60] (the depths the normal-map sweep can meet; below, the quotient is absorbed by the smoothing size it is
// added to).  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o div10 div10.hip ; run: ./div10
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned first, unsigned count, unsigned long long* bad) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long mine = 0;
    for (unsigned k = i; k < count; k += gridDim.x * blockDim.x) {
        const float x = __builtin_bit_cast(float, first + k);
        const float ref = x / 10.0f;
        const float q = x * 0.1f;
        const float r = fmaf(-q, 10.0f, x);
        const float got = fmaf(r, 0.1f, q);
        mine += __builtin_bit_cast(unsigned, ref) != __builtin_bit_cast(unsigned, got);
    }
    if (mine) atomicAdd(bad, mine);
}
int main() {
    unsigned long long* bad;
    hipMalloc(&bad, 8);
    hipMemset(bad, 0, 8);
    const float lo = 0x1p-20f, hi = 0x1p60f;
    const unsigned first = __builtin_bit_cast(unsigned, lo), last = __builtin_bit_cast(unsigned, hi);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, first, last - first + 1, bad);
    unsigned long long h = 0;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("x / 10.0f vs the fma form over %u floats in [2^-20, 2^60]: %llu mismatches\n", last - first + 1, h);
    return h != 0;
}
