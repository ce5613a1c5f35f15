// What does a wave's gather of 64 near-contiguous target records cost the memory pipeline, by record layout?
//   A  12-byte AoS records {I, gx, gy}, one buffer_load_dwordx3 per lane (what the per-pixel pass does)
//   B  16-byte AoS records {I, gx, gy, pad}, one buffer_load_dwordx4 per lane
//   C  three SoA planes, three buffer_load_dword per lane
// Every wave walks its own span of the table in 64-record steps (the near-identity warp: lane i reads record first + i), 4 gathers in
// flight, 1024 or 512 threads x 256 blocks; table 25 MB (Infinity Cache) or 400 MB (HBM).
//   hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip && ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float3v __attribute__((ext_vector_type(3)));
typedef float float4v __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(1024) void k(const float* __restrict__ tab, size_t n_rec, int steps, int shift, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const size_t n_waves = (size_t)gridDim.x * (blockDim.x >> 6);
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const size_t span = n_rec / n_waves;
    const size_t first0 = wave * span;
    const unsigned bytes = (unsigned)((KIND == 1 ? 16u : 12u) * (n_rec < (1u << 27) ? n_rec : (1u << 27)));
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tab), 0, (int)bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tab + n_rec), 0, (int)(4 * n_rec), 0x00020000);
    __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tab + 2 * n_rec), 0, (int)(4 * n_rec), 0x00020000);
    float acc = 0.f;
    for (int s = 0; s < steps; s += 4) {
        float3v a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned idx = (unsigned)(first0 + (size_t)(s + j) * 64 + lane + shift);      // `shift`: records by which the run is off a 64-record boundary
            if (KIND == 0) a[j] = (float3v)__builtin_amdgcn_raw_buffer_load_b96(r, (int)(idx * 12u), 0, 0);
            else if (KIND == 1) { const float4v v = (float4v)__builtin_amdgcn_raw_buffer_load_b128(r, (int)(idx * 16u), 0, 0); a[j] = float3v{v.x, v.y, v.z}; }
            else {
                a[j].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(idx * 4u), 0, 0));
                a[j].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, (int)(idx * 4u), 0, 0));
                a[j].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r2, (int)(idx * 4u), 0, 0));
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += a[j].x + a[j].y * a[j].z;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t n_rec : {(size_t)2 << 20, (size_t)32 << 20}) {
        float* tab; hipMalloc(&tab, n_rec * 16); hipMemset(tab, 0, n_rec * 16);
        for (int threads : {1024, 512})
            for (int kind = 0; kind < 3; ++kind)
                for (int shift : {0, 21}) {
                    const size_t n_waves = (size_t)256 * (threads / 64);
                    const int steps = (int)(n_rec / n_waves / 64) & ~3;
                    float best = 1e9f;
                    for (int rep = 0; rep < 6; ++rep) {
                        hipEventRecord(e0);
                        if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, tab, n_rec, steps, shift, out);
                        else if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, tab, n_rec, steps, shift, out);
                        else hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, tab, n_rec, steps, shift, out);
                        hipEventRecord(e1); hipEventSynchronize(e1);
                        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
                    }
                    const double recs = (double)steps * 64 * n_waves;
                    printf("%s table %4zu MB(12B) threads %4d shift %2d: %.1f us, %.2f T records/s = %.2f TB/s of useful 12-byte records, %.1f ns per wave-gather per CU\n",
                           kind == 0 ? "A dwordx3 AoS12" : kind == 1 ? "B dwordx4 AoS16" : "C 3 x dword SoA", n_rec * 12 >> 20, threads, shift, best * 1e3, recs / (best * 1e-3) * 1e-12,
                           recs * 12 / (best * 1e-3) * 1e-12, best * 1e6 / ((double)steps * (threads / 64)));
                }
        hipFree(tab);
    }
    return 0;
}
