// Waiting for a short stream of kernels without hipStreamSynchronize: the stream's last kernel stores a sequence number into pinned
// host memory (after a system-scope fence), the host spins on it.  tools/ubench/sync_latency.hip: two tiny kernels + a 600-byte
// read-back cost 17.9 us with hipMemcpyAsync + hipStreamSynchronize and 11.9 us when the publishing kernel writes the bytes into the
// pinned buffer itself and the host spins -- 6 us per round trip, which is what the latency-bound paths are made of (one read-back
// per Levenberg-Marquardt evaluation of the pinhole and rig registrations, one per alignment of the spherical one).
// The spin is bounded: after kSpinBudgetUs (2 ms: every alignment of the bench sizes ends sooner) the host falls back to hipStreamSynchronize.
#pragma once
#include "knobs.h"
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdlib>

#include <immintrin.h>

namespace hostwait {

constexpr double kSpinBudgetUs = 2000.0;
// RGBD360_HOST_SPIN_US=<microseconds> overrides the budget; 0 = never spin (wait in hipStreamSynchronize, as a deployment that
// cannot spare a busy core per waiting thread would want)
inline double spin_budget_us() {
    static const double v = [] {
        const char* e = knobs::product("RGBD360_HOST_SPIN_US");
        return e ? atof(e) : kSpinBudgetUs;
    }();
    return v;
}

// flags of every pinned buffer a running kernel publishes into (spin tags, published states / totals / flag words)
constexpr unsigned kPublishedFlags = hipHostMallocCoherent | hipHostMallocMapped;

struct SpinTag {
    unsigned* h = nullptr;      // pinned, device-visible
    unsigned seq = 0;
};
inline hipError_t spin_tag_init(SpinTag* t) {
    if (t->h) return hipSuccess;
    // coherent (fine-grained) + mapped, explicitly: the host polls this word while the stream is still running, which must not depend
    // on the process-wide HIP_HOST_COHERENT default (non-coherent pinned memory is only guaranteed visible at synchronisation points:
    // every wait would burn its whole spin budget first)
    const hipError_t e = hipHostMalloc((void**)&t->h, 64, kPublishedFlags);
    if (e == hipSuccess) *t->h = 0;
    t->seq = 0;
    return e;
}
inline void spin_tag_free(SpinTag* t) {
    if (t->h) (void)hipHostFree(t->h);
    t->h = nullptr;
}

// the last thing a stream does before the host looks: everything earlier in the stream is complete (stream order)
static __global__ void k_tag(unsigned* tag, unsigned seq) {
    __threadfence_system();
    __hip_atomic_store(tag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// n_words 32-bit words from device memory into the pinned host buffer, then the tag
// (one wave: every wave that runs a system-scope fence asks for its own write-back, and a few hundred words do not need four)
static __global__ __launch_bounds__(64) void k_publish(const unsigned* __restrict__ src, unsigned* __restrict__ dst_host, int n_words, unsigned* tag,
                                                unsigned seq) {
    for (int i = threadIdx.x; i < n_words; i += blockDim.x) dst_host[i] = src[i];
    __threadfence_system();
    if (threadIdx.x == 0) __hip_atomic_store(tag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// host side: returns once the tag shows `seq` (spin), or after hipStreamSynchronize when the budget is spent
inline hipError_t wait(SpinTag& t, hipStream_t stream) {
    const double budget = spin_budget_us();
    const auto t0 = std::chrono::steady_clock::now();
    for (int spins = 0; budget > 0.0; ++spins) {
        if (__atomic_load_n(t.h, __ATOMIC_ACQUIRE) == t.seq) return hipSuccess;
        _mm_pause();
        if ((spins & 255) == 255 && std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > budget) break;
    }
    const hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    return __atomic_load_n(t.h, __ATOMIC_ACQUIRE) == t.seq ? hipSuccess : hipErrorUnknown;
}
// enqueue the tag behind whatever the stream holds and wait for it
inline hipError_t tag_and_wait(SpinTag& t, hipStream_t stream) {
    hipLaunchKernelGGL(k_tag, dim3(1), dim3(1), 0, stream, t.h, ++t.seq);
    const hipError_t e = hipGetLastError();
    return e != hipSuccess ? e : wait(t, stream);
}
// copy `bytes` (a multiple of 4) of device memory into a pinned host buffer through the stream and wait for them
inline hipError_t publish_and_wait(SpinTag& t, hipStream_t stream, const void* src_dev, void* dst_host, size_t bytes) {
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, stream, (const unsigned*)src_dev, (unsigned*)dst_host, (int)(bytes / 4), t.h, ++t.seq);
    const hipError_t e = hipGetLastError();
    return e != hipSuccess ? e : wait(t, stream);
}

}  // namespace hostwait
