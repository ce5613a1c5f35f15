// What would one Gauss-Newton iteration of a PERSISTENT coarse-level kernel pay for handing the partial rows of G co-resident
// blocks to every block ("measure a device barrier among a limited co-resident grid")?  Two exchanges, G = 16 .. 256
// blocks of 1024 threads, 2000 iterations each inside one launch, per-iteration time from the wall clock of the launch:
//   counter : thread 0 of a block adds 1 to a device-scope counter behind its row stores, spins until it shows (it + 1) * G, the block
//             then loads all G rows (what a grid barrier + the fused launch's stage_pending would do)
//   tagged  : a row's 32 values travel as 16-byte {value, generation} elements; every thread polls the elements it sums until they
//             carry the generation -- one round trip, no counter, no wait for the stores' acknowledgement
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/grid_exchange tools/ubench/grid_exchange.hip && /tmp/grid_exchange
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

constexpr int kThreads = 1024, kVals = 32, kMaxRows = 256;
constexpr long kSpinCap = 1 << 22;      // a block that is not co-resident with the others ends the run instead of hanging the GPU

struct alignas(16) Elem { double v; unsigned long long gen; };

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Elem load_elem(const Elem* p) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    return __builtin_bit_cast(Elem, r);
}
__device__ __forceinline__ void store_elem(Elem* p, Elem e) {
    const u32x4 r = __builtin_bit_cast(u32x4, e);
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(r) : "memory");
}

__global__ __launch_bounds__(kThreads) void k_counter(double* rows /*[2][kMaxRows][kVals]*/, unsigned* ctr, int G, int iters, double* out, int* bad) {
    const int tid = threadIdx.x, b = blockIdx.x;
    const int v = tid % kVals, q = tid / kVals;      // 32 groups of 32
    __shared__ double red[kThreads];
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        double* R = rows + (size_t)(it & 1) * kMaxRows * kVals;
        if (tid < kVals) __hip_atomic_store(&R[b * kVals + tid], (double)(b + it) + acc * 1e-30, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            long spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it + 1) * (unsigned)G)
                if (++spins > kSpinCap) { *bad = 1; break; }
        }
        __syncthreads();
        if (*(volatile int*)bad) return;
        double s = 0.0;
        for (int r = q; r < G; r += kThreads / kVals) s += __hip_atomic_load(&R[r * kVals + v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        red[tid] = s;
        __syncthreads();
        if (tid < kVals) {
            double t = 0.0;
            for (int g = 0; g < kThreads / kVals; ++g) t += red[g * kVals + tid];
            acc = t;
        }
        __syncthreads();
    }
    if (tid < kVals && b == 0) out[tid] = acc;
}

__global__ __launch_bounds__(kThreads) void k_tagged(Elem* rows /*[2][kMaxRows][kVals]*/, int G, int iters, double* out, int* bad, unsigned long long gen0) {
    const int tid = threadIdx.x, b = blockIdx.x;
    const int v = tid % kVals, q = tid / kVals;
    __shared__ double red[kThreads];
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        Elem* R = rows + (size_t)(it & 1) * kMaxRows * kVals;
        const unsigned long long gen = gen0 + it + 1;
        if (tid < kVals) store_elem(&R[b * kVals + tid], Elem{(double)(b + it) + acc * 1e-30, gen});
        double s = 0.0;
        for (int r = q; r < G; r += kThreads / kVals) {
            Elem e = load_elem(&R[r * kVals + v]);
            long spins = 0;
            while (e.gen != gen) {
                if (++spins > kSpinCap) { *bad = 1; break; }
                __builtin_amdgcn_s_sleep(1);
                e = load_elem(&R[r * kVals + v]);
            }
            s += e.v;
        }
        red[tid] = s;
        __syncthreads();
        if (*(volatile int*)bad) return;
        if (tid < kVals) {
            double t = 0.0;
            for (int g = 0; g < kThreads / kVals; ++g) t += red[g * kVals + tid];
            acc = t;
        }
        __syncthreads();
    }
    if (tid < kVals && b == 0) out[tid] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    double* rows; Elem* erows; unsigned* ctr; double* out; int* bad;
    CK(hipMalloc(&rows, 2 * kMaxRows * kVals * sizeof(double)));
    CK(hipMalloc(&erows, 2 * kMaxRows * kVals * sizeof(Elem)));
    CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&out, kVals * 8)); CK(hipMalloc(&bad, 4));
    CK(hipMemset(erows, 0, 2 * kMaxRows * kVals * sizeof(Elem)));
    CK(hipMemset(bad, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    unsigned long long gen0 = 0;
    for (int G : {16, 32, 64, 128, 256}) {
        float ms_c = 0, ms_t = 0; int hbad = 0; double h[2] = {0, 0};
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(ctr, 0, 4));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_counter, dim3(G), dim3(kThreads), 0, 0, rows, ctr, G, iters, out, bad);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_c, e0, e1));
        }
        CK(hipMemcpy(&h[0], out, 8, hipMemcpyDeviceToHost));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_tagged, dim3(G), dim3(kThreads), 0, 0, erows, G, iters, out, bad, gen0);
            gen0 += iters;
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_t, e0, e1));
        }
        CK(hipMemcpy(&h[1], out, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
        printf("G %3d blocks: counter barrier + row loads %6.2f us / iteration, tagged rows %6.2f us / iteration   (sums %.1f %.1f%s)\n", G,
               ms_c * 1e3 / iters, ms_t * 1e3 / iters, h[0], h[1], hbad ? "  SPIN CAP HIT" : "");
        if (hbad) return 2;
    }
    return 0;
}
