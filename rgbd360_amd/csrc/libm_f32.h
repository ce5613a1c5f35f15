// asinf / atanf / atan2f as glibc 2.35's libm.so.6 computes them on x86-64 (sysdeps/ieee754/flt-32: e_asinf.c, s_atanf.c, e_atan2f.c -- the
// fdlibm float code, compiled without fused multiply-adds), restated operation for operation so that the device can reproduce the
// REFERENCE's warp arithmetic bit for bit (RegisterPhotoICP.h:2674-2680 calls asin / atan2 on floats).  Neither function is correctly
// rounded (asinf differs from the rounded float64 value on 0.2 % of its inputs, atan2f on 13 %), so "the same index as the reference"
// means THIS operation sequence: every product, sum and quotient below is one IEEE float32 operation, in the order of the library's
// machine code (constants read out of its .rodata).  Proven equal to the host's libm -- asinf on every float of [-1, 1], atanf on every
// float, atan2f on 4e9 pairs -- by tools/libm_f32_check.cpp (host compile of this header) and rgbd360_selftest_libm (device compile).
// Requires: no floating-point contraction (the library is built with -ffp-contract=off) and correctly rounded float division and square
// root (hipcc's default, -fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define LIBM32_HD __host__ __device__ __forceinline__
#else
#define LIBM32_HD inline
#endif

namespace libm32 {

LIBM32_HD uint32_t f2u(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
LIBM32_HD float u2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
LIBM32_HD float fabs32(float x) { return u2f(f2u(x) & 0x7fffffffu); }
// Correctly rounded on both sides: the host's sqrtss / divss, and on the device the math library's sqrtf and operator/ under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt (rgbd360_selftest_libm is the check: asinf's |x| >= 0.5 branch goes through both).
LIBM32_HD float sqrt32(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return sqrtf(x);
#else
    return __builtin_sqrtf(x);
#endif
}
LIBM32_HD float div32(float a, float b) { return a / b; }

// e_asinf.c
LIBM32_HD float asinf_(float x) {
    const float pio2_hi = u2f(0x3fc90fdbu), pio2_lo = u2f(0xb33bbd2eu), pio4_hi = u2f(0x3f490fdbu);
    const float p0 = u2f(0x3e2aaae4u), p1 = u2f(0x3d9980f2u), p2 = u2f(0x3d3a3f25u), p3 = u2f(0x3cc6141eu), p4 = u2f(0x3d2cb694u);
    const uint32_t hx = f2u(x), ix = hx & 0x7fffffffu;
    if (ix == 0x3f800000u) return x * pio2_hi + x * pio2_lo;      // asin(+-1) = +-pi/2 with inexact
    if (ix > 0x3f800000u) return (x - x) / (x - x);                // |x| > 1: NaN
    if (ix < 0x3f000000u) {                                        // |x| < 0.5
        if (ix < 0x32000000u) return x;                            // |x| < 2^-27
        const float t = x * x;
        float w = p4 * t;
        w = w + p3; w = w * t;
        w = w + p2; w = w * t;
        w = w + p1; w = w * t;
        w = w + p0; w = w * t;
        return x + w * x;
    }
    // 1 > |x| >= 0.5
    float w = 1.0f - fabs32(x);
    const float t = w * 0.5f;
    float p = p4 * t;
    p = p + p3; p = p * t;
    p = p + p2; p = p * t;
    p = p + p1; p = p * t;
    p = p + p0; p = p * t;
    const float s = sqrt32(t);
    float r;
    if (ix >= 0x3f79999au) {                                       // |x| > 0.975
        float a = p * s;
        a = a + s;
        a = a + a;
        r = pio2_hi - (a - pio2_lo);
    } else {
        w = u2f(f2u(s) & 0xfffff000u);
        const float c = div32(t - w * w, s + w);
        const float pp = (s + s) * p - (pio2_lo - (c + c));
        const float q = pio4_hi - (w + w);
        r = pio4_hi - (pp - q);
    }
    return (int32_t)hx > 0 ? r : -r;
}

// s_atanf.c
LIBM32_HD float atanf_(float x) {
    const float aT0 = u2f(0x3eaaaaabu), aT2 = u2f(0x3e124925u), aT4 = u2f(0x3dba2e6eu), aT6 = u2f(0x3d886b35u), aT8 = u2f(0x3d4bda59u),
                aT10 = u2f(0x3c8569d7u);
    // the odd coefficients are negative; the library holds aT9 and the magnitudes of aT7 .. aT1, which it subtracts
    const float aT9 = u2f(0xbd15a221u), m7 = u2f(0x3d6ef16bu), m5 = u2f(0x3d9d8795u), m3 = u2f(0x3de38e38u), m1 = u2f(0x3e4ccccdu);
    const uint32_t hx = f2u(x), ix = hx & 0x7fffffffu;
    float hi, lo;
    int id;
    if (ix >= 0x4c000000u) {                                       // |x| >= 2^25
        if (ix > 0x7f800000u) return x + x;                        // NaN
        if ((int32_t)hx > 0) return u2f(0x33a22168u) + u2f(0x3fc90fdau);
        return u2f(0xbfc90fdau) - u2f(0x33a22168u);
    }
    if (ix < 0x3ee00000u) {                                        // |x| < 0.4375
        if (ix < 0x31000000u) return x;                            // |x| < 2^-29
        id = -1;
        hi = lo = 0.f;
    } else {
        const float ax = fabs32(x);
        if (ix < 0x3f980000u) {                                    // |x| < 1.1875
            if (ix < 0x3f300000u) {                                // 7/16 <= |x| < 11/16
                id = 0;
                x = div32((ax + ax) - 1.0f, ax + 2.0f);
                hi = u2f(0x3eed6338u); lo = u2f(0x31ac3769u);
            } else {                                               // 11/16 <= |x| < 19/16
                id = 1;
                x = div32(ax - 1.0f, ax + 1.0f);
                hi = u2f(0x3f490fdau); lo = u2f(0x33222168u);
            }
        } else if (ix < 0x401c0000u) {                             // |x| < 2.4375
            id = 2;
            x = div32(ax - 1.5f, ax * 1.5f + 1.0f);
            hi = u2f(0x3f7b985eu); lo = u2f(0x33140fb4u);
        } else {                                                   // 2.4375 <= |x| < 2^25
            id = 3;
            x = div32(-1.0f, ax);
            hi = u2f(0x3fc90fdau); lo = u2f(0x33a22168u);
        }
    }
    const float z = x * x;
    const float w = z * z;
    float s1 = aT10 * w;
    s1 = s1 + aT8; s1 = s1 * w;
    s1 = s1 + aT6; s1 = s1 * w;
    s1 = s1 + aT4; s1 = s1 * w;
    s1 = s1 + aT2; s1 = s1 * w;
    s1 = s1 + aT0; s1 = s1 * z;
    float s2 = aT9 * w;
    s2 = s2 - m7; s2 = s2 * w;
    s2 = s2 - m5; s2 = s2 * w;
    s2 = s2 - m3; s2 = s2 * w;
    s2 = s2 - m1; s2 = s2 * w;
    const float xs = (s1 + s2) * x;
    if (id < 0) return x - xs;
    const float r = hi - ((xs - lo) - x);
    return (int32_t)hx < 0 ? -r : r;
}

// e_atan2f.c (finite and non-finite arguments alike)
LIBM32_HD float atan2f_(float y, float x) {
    const float tiny = u2f(0x0da24260u), pi = u2f(0x40490fdbu), pio2 = u2f(0x3fc90fdbu), pio4 = u2f(0x3f490fdbu);
    const float neg_pi_lo = u2f(0x33bbbd2eu);                      // pi_lo = -8.7422776573e-08
    const uint32_t hx = f2u(x), hy = f2u(y), ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    if (ix > 0x7f800000u || iy > 0x7f800000u) return x + y;        // NaN
    if (hx == 0x3f800000u) return atanf_(y);                       // x = 1
    const int m = (int)((hy >> 31) & 1u) | (int)((hx >> 30) & 2u);  // 2 sign(x) + sign(y)
    if (iy == 0) {
        if (m < 2) return y;                                       // atan(+-0, +anything) = +-0
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if (ix == 0) return (int32_t)hy < 0 ? -pio2 - tiny : pio2 + tiny;
    if (ix == 0x7f800000u) {
        if (iy == 0x7f800000u) {
            switch (m) {
                case 0: return pio4 + tiny;
                case 1: return -pio4 - tiny;
                case 2: return 3.0f * pio4 + tiny;
                default: return -3.0f * pio4 - tiny;
            }
        }
        switch (m) {
            case 0: return 0.0f;
            case 1: return -0.0f;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if (iy == 0x7f800000u) return (int32_t)hy < 0 ? -pio2 - tiny : pio2 + tiny;
    const int32_t d = (int32_t)iy - (int32_t)ix;
    const int k = d >> 23;
    float z;
    if (d > 0x1e7fffff) z = pio2 - u2f(0x333bbd2eu);               // |y / x| > 2^60: pi/2 + 0.5 pi_lo
    else if ((int32_t)hx < 0 && k < -60) z = 0.0f;                 // |y| / x < -2^60
    else z = atanf_(fabs32(div32(y, x)));
    switch (m) {
        case 0: return z;
        case 1: return u2f(f2u(z) ^ 0x80000000u);
        case 2: return pi - (z + neg_pi_lo);
        default: return (z + neg_pi_lo) - pi;
    }
}

// roundf: half away from zero (RegisterPhotoICP.h:2679-2680 rounds the scaled angles with round())
LIBM32_HD float roundf_(float x) {
    const uint32_t hx = f2u(x), ix = hx & 0x7fffffffu;
    if (ix >= 0x4b000000u) return x;                               // |x| >= 2^23 (or not finite): an integer already
    const float ax = u2f(ix);
    float t = (float)(int32_t)ax;                                  // truncation: exact below 2^23
    if (ax - t >= 0.5f) t = t + 1.0f;                              // (ax - t is exact)
    return u2f(f2u(t) | (hx & 0x80000000u));
}

}  // namespace libm32
