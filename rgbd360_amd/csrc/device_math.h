// device_math.h -- the few device helpers both device translation units use (rgbd360_api.hip through photo_icp_kernels.h,
// rgbd360_frame360.hip through frame360_kernels.h): the correctly rounded float32 square root / reciprocal the parity definition rests
// on (proven by exhaustion: rgbd360_selftest_math), the 24-bit division, the sphere-cloud point of the three conventions.
#pragma once
#include <hip/hip_runtime.h>

namespace r360 {

constexpr double kPI = 3.14159265359;           // Miscellaneous.h:44 (truncated literal, double)

// Correctly rounded sqrt for x == 0 and normal finite x in [2^-60, 2^60]: reciprocal-square-root estimate (<= 1 ulp) and
// ONE coupled Newton step, s = x y, s += (x - s s) (y / 2), with the residual taken exactly by an fma.  Proven by
// exhaustion, not by analysis: rgbd360_selftest_math (and tools/ubench/rn_variants.hip) compare it with the compiler's IEEE
// sqrtf for every float of that range -- 0 mismatches on gfx950.  (The first version corrected the hardware sqrt with two
// +-1 ulp residual tests: 9 instructions instead of 6.)
__device__ __forceinline__ float sqrt_rn(float x, float& y) {           // y: the hardware estimate of 1 / sqrt(x) it starts from (<= 1 ulp)
    y = __builtin_amdgcn_rsqf(fmaxf(x, 1.17549435e-38f));      // the clamp only matters for x == 0: s = 0 * y = 0
    const float s = x * y;
    const float h = 0.5f * y;
    const float r = fmaf(-s, s, x);
    return fmaf(r, h, s);
}
__device__ __forceinline__ float sqrt_rn(float x) {
    float y;
    return sqrt_rn(x, y);
}
// Correctly rounded 1/x for normal finite |x| in [2^-60, 2^60]: hardware estimate (<= 1 ulp) + one Newton step with an exact
// fma residual; exhaustively equal to the IEEE quotient 1.f / x on gfx950 (same self-test).
__device__ __forceinline__ float rcp_rn(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    const float e = fmaf(-x, r, 1.f);
    return fmaf(e, r, r);
}

// n / d and n % d for 0 <= n < 2^24, d >= 1 (images are < 16 Mpx): a float estimate of the quotient, exact after one
// correction either way -- a runtime-divisor integer division costs ~30 VALU instructions, this one ~8.
__device__ __forceinline__ void divmod24(int n, int d, int& q, int& rem) {
    q = (int)((float)n * (1.0f / (float)d));
    rem = n - q * d;
    if (rem < 0) { rem += d; --q; }
    else if (rem >= d) { rem -= d; ++q; }
}

// Frame360 sphere clouds (Frame360.h:555-612, Frame360_stereo.h:454-512) and the RegisterPhotoICP convention.
__device__ __forceinline__ void sphere_point(int convention, float d, float sp, float cp, float st, float ct, float& x, float& y, float& z) {
    const float qnan = __builtin_nanf("");
    x = qnan; y = qnan; z = qnan;
    if (convention == 0) {
        if (d != 0) {
            x = sp * d;
            y = -cp * st * d;
            z = -cp * ct * d;
        }
    } else if (convention == 1) {
        if (d > 0.f && d < 15.f) {
            x = st * cp * d;
            y = sp * d;
            z = ct * cp * d;
        }
    } else {
        if (d != 0) {
            x = d * sp;
            y = -d * cp * st;
            z = -d * cp * ct;
        }
    }
}

}  // namespace r360
