// Host check of rgbd360_amd/csrc/libm_f32.h against the C library this box links (the reference's asinf / atanf / atan2f / roundf):
// asinf on every float of [-1, 1] and beyond, atanf and roundf on every float, atan2f on N random pairs (all binades of the warp's operand
// range, every sign combination, zeros) -- bit for bit.
//   g++ -O2 -ffp-contract=off -pthread -o /tmp/libm_f32_check tools/libm_f32_check.cpp && /tmp/libm_f32_check [pairs_in_millions [stride]]
//   (stride > 1: every stride-th float only -- the CPU test suite's quick form, tests/test_libm_restatement.py)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <thread>
#include <vector>
#include "../rgbd360_amd/csrc/libm_f32.h"

static bool same(float a, float b) {
    const uint32_t x = libm32::f2u(a), y = libm32::f2u(b);
    return x == y || ((x & 0x7fffffffu) > 0x7f800000u && (y & 0x7fffffffu) > 0x7f800000u);      // any NaN equals any NaN
}
int main(int argc, char** argv) {
    const unsigned T = std::max(1u, std::thread::hardware_concurrency());
    const long long pairs = (argc > 1 ? atoll(argv[1]) : 2000) * 1000000LL;
    const uint64_t stride = argc > 2 ? (uint64_t)atoll(argv[2]) : 1;
    std::atomic<unsigned long long> bad_asin{0}, bad_atan{0}, bad_round{0}, bad_atan2{0};
    std::atomic<uint32_t> first_asin{0}, first_atan{0}, first_round{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            unsigned long long ba = 0, bt = 0, br = 0;
            for (uint64_t u = t * stride; u < (1ull << 32); u += T * stride) {
                const float x = libm32::u2f((uint32_t)u);
                const uint32_t ix = (uint32_t)u & 0x7fffffffu;
                if (ix <= 0x3fc00000u && !same(libm32::asinf_(x), asinf(x))) {      // |x| <= 1.5: the NaN branch too
                    if (!ba) first_asin = (uint32_t)u;
                    ++ba;
                }
                if (!same(libm32::atanf_(x), atanf(x))) {
                    if (!bt) first_atan = (uint32_t)u;
                    ++bt;
                }
                if (!same(libm32::roundf_(x), roundf(x))) {
                    if (!br) first_round = (uint32_t)u;
                    ++br;
                }
            }
            bad_asin += ba; bad_atan += bt; bad_round += br;
            // atan2f: exponents 2^-40 .. 2^40 around 1 (the warp's operands are metres), a share of exact zeros, ones, equal magnitudes
            unsigned long long b2 = 0;
            uint64_t s = 0x9E3779B97F4A7C15ull * (t + 1);
            auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
            for (long long i = t; i < pairs; i += T) {
                const uint64_t r = rnd(), q = rnd();
                uint32_t a = (uint32_t)r, b = (uint32_t)q;
                const int ea = 87 + (int)((r >> 32) % 81), eb = 87 + (int)((q >> 32) % 81);
                a = (a & 0x807fffffu) | ((uint32_t)ea << 23);
                b = (b & 0x807fffffu) | ((uint32_t)eb << 23);
                const unsigned kind = (unsigned)(r >> 40) & 63u;
                if (kind == 0) a &= 0x80000000u;                   // y = +-0
                if (kind == 1) b &= 0x80000000u;                   // x = +-0
                if (kind == 2) b = 0x3f800000u;                    // x = 1
                if (kind == 3) b = (b & 0x80000000u) | (a & 0x7fffffffu);      // |x| = |y|
                if (kind == 4) b = (b & 0x807fffffu) | (a & 0x7f800000u);      // same binade
                const float y = libm32::u2f(a), x = libm32::u2f(b);
                if (!same(libm32::atan2f_(y, x), atan2f(y, x))) ++b2;
            }
            bad_atan2 += b2;
        });
    for (auto& x : th) x.join();
    printf("asinf  on every float with |x| <= 1.5: %llu differ (first 0x%08x)\n", (unsigned long long)bad_asin, (unsigned)first_asin);
    printf("atanf  on every float:                 %llu differ (first 0x%08x)\n", (unsigned long long)bad_atan, (unsigned)first_atan);
    printf("roundf on every float:                 %llu differ (first 0x%08x)\n", (unsigned long long)bad_round, (unsigned)first_round);
    printf("atan2f on %lld random pairs:    %llu differ\n", pairs, (unsigned long long)bad_atan2);
    return (bad_asin | bad_atan | bad_round | bad_atan2) ? 1 : 0;
}
