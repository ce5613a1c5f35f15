// How many HIP streams of one process really run concurrently?  Each stream gets one single-block kernel that spins for a fixed
// wall-clock time (bounded: it always exits); total time / spin time = number of serialised groups = streams per hardware queue.
// hipcc --offload-arch=gfx950 -O2 -o tools/ubench/stream_queues tools/ubench/stream_queues.hip
//   ./stream_queues [mode]   mode 0: plain streams, 1: alternate priorities (normal/high/low)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void spin(long long ticks, int* out) {
    const long long t0 = wall_clock64();
    long long t = t0;
    int guard = 0;
    while (t - t0 < ticks && guard < (1 << 26)) { t = wall_clock64(); ++guard; }
    if (out) *out = guard;
}
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    printf("priority range: least %d greatest %d; mode %d\n", lo, hi, mode);
    const long long ticks = 100000 * 2;          // wall_clock64 runs at 100 MHz: 2 ms
    int* d = nullptr;
    hipMalloc(&d, 4);
    for (int n = 1; n <= 12; ++n) {
        std::vector<hipStream_t> ss(n);
        for (int i = 0; i < n; ++i) {
            int prio = 0;
            if (mode == 1) prio = (i % 3 == 0) ? 0 : (i % 3 == 1 ? hi : lo);
            if (mode == 2) prio = (i < 3) ? 0 : (i < 6 ? hi : lo);
            hipStreamCreateWithPriority(&ss[i], hipStreamNonBlocking, prio);
        }
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, ss[i], 1000, (int*)nullptr);
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, ss[i], ticks, (int*)nullptr);
        hipDeviceSynchronize();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("%2d streams: %.2f ms  (~%.1f serialised groups)\n", n, ms, ms / 2.0);
        for (auto s : ss) hipStreamDestroy(s);
    }
    return 0;
}
