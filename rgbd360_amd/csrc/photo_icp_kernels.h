// photo_icp_kernels.h -- gfx950 (MI355X, wave64) device code of the dense spherical alignment path.
//
// Kernels and the reference loops they replace ("RPI.h" = include/RegisterPhotoICP.h of EduFdez/rgbd360):
//   k_gray_u8            cv::cvtColor(CV_RGB2GRAY)+convertTo(1/255)            RPI.h:485-486, 502-503
//   k_depth_to_f32       convertTo(CV_32FC1, 0.001)                             RPI.h:316-319
//   k_pyrdown_gray       buildPyramid -> cv::pyrDown                            RPI.h:292-308
//   k_pyrdown_depth      buildPyramidRange                                      RPI.h:312-354
//   k_gradient_rec       calcGradientXY + seam mask, writes {v,gx,gy} records   RPI.h:365-398, 4538-4549
//   k_src_rec            LUT_xyz_sphere build, writes {x,y,z,Isrc} records      RPI.h:4554-4587
//   k_eval<METHOD,HG>    errorPhotoICP_sphere + calcHessGrad_sphere, fused      RPI.h:2545-2739, 2745-3228
//   k_solve              reductions' tail + the serial part of alignFrames360   RPI.h:4599-4722
//
// This translation unit is compiled with -ffp-contract=off: the warp front end (rotation, norm, asin, atan2,
// rounding) must produce the same float32 values as the CPU oracle so that the nearest-neighbour target pixel
// is identical; contraction is re-enabled locally (clang fp contract) for the Jacobian / normal-equation part,
// whose values only need to agree to float32 rounding.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "device_math.h"
#include "gn_math.h"
#include "libm_f32.h"

namespace r360 {

constexpr float  kInvalidPoint = -10000.f;      // RPI.h:40
constexpr int    kEvalThreads = 1024;
constexpr int    kNumPartials = 32;             // doubles per block partial
// partial slots
enum { P_H = 0 /*21*/, P_G = 21 /*6*/, P_E2P = 27, P_E2D = 28, P_NP = 29, P_ND = 30, P_NVIS = 31 };

struct F3 {
    float a, b, c;
};

struct LevelDev {
    int rows, cols, n;
    float half_nRows;        // 0.5*nRows - 0.5
    float angle_res_inv;     // 1 / float(2*PI/nCols)
    float pi_k;              // float(PI * angle_res_inv): the +PI of theta_trg (RPI.h:2677) folded into the column scaling
    const float4* src;       // {x, y, z, Isrc} per source pixel; x == -10000 marks an invalid point
    const F3* trgP;          // {Itrg, gradX, gradY} per target pixel
    const F3* trgD;          // {Dtrg, dgradX, dgradY} per target pixel
    // Recompute forms of the source stream (SRC = 1, 2 of the per-pixel pass; SURVEY.md 8d "recompute-from-depth variant": 8 instead of
    // 16 bytes per source pixel): the point is re-formed per pixel as LUT_xyz_sphere does (RPI.h:4573-4582, bit for bit) from the
    // pixel's depth and the angle tables.  SRC 1 reads depth / intensity from the level's float planes, SRC 2 from {depth, Isrc} records.
    const float *depth_src = nullptr, *gray_src = nullptr;
    const float2* src2 = nullptr;                            // {depth, Isrc} per source pixel (lock-step engine)
    const float2 *tabT = nullptr, *tabP = nullptr;           // {sin, cos} of theta per column / of phi per row (host libm, RPI.h:4556-4571)
    float min_depth = 0.f, max_depth = 0.f;
    int libm = 0;            // 1: the warp in the REFERENCE's arithmetic (libm_f32.h), rgbd360_set_index_arithmetic
};

struct EvalConsts {
    float  sigma_photo, sigma_depth, thr_photo, thr_depth;
    float  sigma_photo_inv_f;   // float(1./stdDevPhoto)   RPI.h:2774
    double sigma_photo_inv_d;   // 1./stdDevPhoto          RPI.h:2561
};

// Device-resident state of one alignment (one per context).  Poses/H column-major.
struct GNState {
    float  cand[16];     // pose the next / last fused pass is evaluated at (first: read by every k_eval block)
    int    done;         // level finished (or status != 0): later launches of this level exit immediately
    int    level_active; // pyramid level being optimised; launches tagged with another level are no-ops
    int    it, status, first, n_evals;
    int    pend_nb, pend_npix;   // fused-solve schedule (k_eval_fs): block rows / pixels of a pass at `cand` on level `level_active`
                                 // whose partial rows have not been reduced yet (0: nothing pending)
    int    iters[8];     // accepted iterations per level (num_iterations, RPI.h:177)
    float  pose[16];     // accepted pose of the current level
    float  H[36], g[6];  // normal equations at `pose`
    float  Hused[36], gused[6];  // those of the last Gauss-Newton step actually taken (= reference `hessian`)
    float  update[6];
    double lambda;
    double error, new_error, diff_error;
    double tot[kNumPartials];        // reduced partials of the last pass
    double acc_e2p, acc_e2d;         // error sums at the accepted pose
    long long acc_np, acc_nd, acc_nvis, used_nvis, used_npix;
    unsigned long long stamps[8];    // diagnostic build only (RGBD360_SOLVE_STAMPS): 100 MHz timestamps of k_solve phases
};

struct SolveCfg {
    int    level;         // pyramid level this launch belongs to
    int    mode;          // 0 Gauss-Newton logic, 1 reduce only
    int    forced;        // apply every step, never terminate
    int    max_iters;
    int    n_pixels;
    int    occ;           // 0: error = sqrt(sum / n) (RPI.h:2738); 1/2: avPhotoResidual + avDepthResidual (RPI.h:3358-3366, 3848-3855)
    double tol_residual, tol_update;
    // when set, the state is also written into this pinned host buffer and host_seq is stored into *host_tag behind it (system
    // scope): the latency-bound callers read their result without a copy launch and without hipStreamSynchronize (host_wait.h)
    GNState* host_state = nullptr;
    unsigned* host_tag = nullptr;
    unsigned host_seq = 0;
};

// ---------------------------------------------------------------------------------------------------------
// scalar helpers.  The warp front end below is the *device arithmetic definition* of the pixel warp: a fixed
// sequence of IEEE-754 basic operations (fma, mul, add, correctly rounded sqrt and reciprocal) that the CPU oracle
// repeats operation for operation in its math_mode 1, so warped pixel indices agree bit for bit.
// ---------------------------------------------------------------------------------------------------------
// round-half-up to the nearest integer, floor(x + 0.5) with the sum taken exactly: one v_cvt_rpi_i32_f32.  Equals
// C round() except at exact negative ties, which cannot change a visible pixel index (negative rows / columns are
// dropped).  rgbd360_selftest_math checks the instruction against floor((double)x + 0.5) over the index range.
__device__ __forceinline__ int round_index(float x) {
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// atan(t) for t in [0, 1(+ulps)]: odd minimax polynomial, 9 coefficients, relative error 1.3e-8.
// What has been measured against it (none shipped; the pass is bound by vector issue AND the memory round trip of each step together --
// DESIGN.md 3.1 -- and neither lever moved the launch alone):
//   round 4: the quadrant angle T(|y| / (|y| + |x|)) from a 256-interval quadratic table in LDS instead of min / max ratio + polynomial +
//            reflection (8.5 instead of 16 vector instructions per angle, 116 instead of 144 per pixel): no change at any size;
//   round 6: both angles' Horner chains in lock step on the packed pipe (v_pk_fma_f32, bit-identical results) and the 27 accumulations
//            as 15 packed ones (115 vector instructions per pixel instead of 135, 26 of them packed): 14.42 -> 14.54 us, the packed
//            instruction costs 1.47 issue slots (profiles/r03_ubench_valu.txt) and the chains' depth is unchanged
//            (profiles/r06_headline_kernel_experiment.txt, the diff: profiles/r06_pk_math_experiment.patch).
__device__ __forceinline__ float atan_unit(float t) {
    const float q0 = -0.3333333195069166f, q1 = 0.19999765993465415f, q2 = -0.14279110844310372f,
                q3 = 0.11037993832882714f, q4 = -0.08673169371217875f, q5 = 0.06284358078457526f,
                q6 = -0.03627014369584507f, q7 = 0.01375026672953864f, q8 = -0.00244702708829393f;
    const float s = t * t;
    float p = fmaf(s, q8, q7);
    p = fmaf(s, p, q6);
    p = fmaf(s, p, q5);
    p = fmaf(s, p, q4);
    p = fmaf(s, p, q3);
    p = fmaf(s, p, q2);
    p = fmaf(s, p, q1);
    p = fmaf(s, p, q0);
    return fmaf(t * s, p, t);
}
// atan2(y, x) from t = min(|y|,|x|) / max(|y|,|x|) supplied by the caller (both angles share one reciprocal).
__device__ __forceinline__ float atan2_from_t(float y, float x, float ay, float ax, float t) {
    float a = atan_unit(t);
    if (ay > ax) a = 1.57079637f - a;
    if (__builtin_signbit(x)) a = 3.14159274f - a;
    return copysignf(a, y);
}

struct PoseRT {
    float r00, r01, r02, r10, r11, r12, r20, r21, r22, tx, ty, tz;
};
__device__ __forceinline__ PoseRT load_pose(const float* P) {   // column-major 4x4
    PoseRT T;
    T.r00 = P[0]; T.r10 = P[1]; T.r20 = P[2];
    T.r01 = P[4]; T.r11 = P[5]; T.r21 = P[6];
    T.r02 = P[8]; T.r12 = P[9]; T.r22 = P[10];
    T.tx = P[12]; T.ty = P[13]; T.tz = P[14];
    return T;
}

// Shared front end of RPI.h:2663-2684 / 2959-2989.  Returns the target pixel index (valid only when `vis`).
// Device arithmetic definition (the oracle's math_mode 1 repeats it operation for operation):
//   p' = R p + t with fused multiply-adds;  rho^2 = Y^2 + Z^2,  d^2 = X^2 + rho^2
//   phi   = atan2(X, rho)      (= asin(X/d) of the reference, RPI.h:2676)      rho = correctly rounded sqrt
//   theta = atan2(Y, Z);  column = round(theta * k + PI*k)   (RPI.h:2677-2680 with the +PI folded into the scaling)
//   both quotients min/max come from ONE correctly rounded reciprocal r = 1 / (mx_phi * mx_theta)
// Every step is an IEEE-754 basic operation, so x86 and gfx950 agree bit for bit on the pixel index.
struct WarpConsts {        // per-lane copies (VGPRs) of the wave-uniform addends: a VOP3 fma reads one scalar only
    float tx, ty, tz, half_nRows, pi_k;
};
__device__ __forceinline__ WarpConsts make_warp_consts(const PoseRT& T, const LevelDev& lv) {
    WarpConsts c = {T.tx, T.ty, T.tz, lv.half_nRows, lv.pi_k};
    asm volatile("" : "+v"(c.tx), "+v"(c.ty), "+v"(c.tz), "+v"(c.half_nRows), "+v"(c.pi_k));
    return c;
}
// The same front end in the REFERENCE's arithmetic (rgbd360_set_index_arithmetic(ctx, 1); the oracle's math_mode 0 is this sequence on the
// CPU, with the C library's functions): Eigen's fixed-size product without fused multiply-adds, norm(), 1 / dist, asinf, atan2f + the
// double PI, roundf -- RPI.h:2663-2684 as compiled on the reference's platform, asinf / atan2f restated operation for operation
// (libm_f32.h).  Indices, visibility and d^2 are bit-equal to the reference's; ~110 instructions instead of 40, so the fast
// definition above stays the default.
__device__ __forceinline__ void warp_pixel_libm(const PoseRT& T, const WarpConsts& wc, float px, float py, float pz, const LevelDev& lv,
                                             float& X, float& Y, float& Z, float& d2, int& tr, int& tc) {
    X = ((T.r00 * px + T.r01 * py) + T.r02 * pz) + wc.tx;
    Y = ((T.r10 * px + T.r11 * py) + T.r12 * pz) + wc.ty;
    Z = ((T.r20 * px + T.r21 * py) + T.r22 * pz) + wc.tz;
    d2 = (X * X + Y * Y) + Z * Z;
    const float dist = libm32::sqrt32(d2);
    const float dist_inv = libm32::div32(1.f, dist);
    const float phi = libm32::asinf_(X * dist_inv);
    const float theta = (float)((double)libm32::atan2f_(Y, Z) + r360::kPI);
    const float fr = libm32::roundf_(lv.half_nRows - phi * lv.angle_res_inv);
    const float fc = libm32::roundf_(theta * lv.angle_res_inv);
    // (int) of a float as x86 converts it: out of range or NaN -> INT_MIN (never a visible index)
    tr = (fr >= -2147483648.f && fr < 2147483648.f) ? (int)fr : (int)0x80000000;
    tc = (fc >= -2147483648.f && fc < 2147483648.f) ? (int)fc : (int)0x80000000;
}
__device__ __forceinline__ void warp_pixel_rc(const PoseRT& T, const WarpConsts& wc, float px, float py, float pz,
                                              const LevelDev& lv, float& X, float& Y, float& Z, float& rho2, float& d2,
                                              int& tr, int& tc, unsigned long long& vis_mask, float& inv_rho) {
#ifndef RGBD360_NO_LIBM_WARP                          // (A/B builds of the default path without the branch: tools/ab_libs.py)
    if (lv.libm) {                                    // uniform (a kernel argument)
        warp_pixel_libm(T, wc, px, py, pz, lv, X, Y, Z, d2, tr, tc);
        rho2 = fmaf(Z, Z, Y * Y);                     // float32 data of the Jacobian, like inv_rho
        (void)sqrt_rn(rho2, inv_rho);
        vis_mask = __builtin_amdgcn_ballot_w64((unsigned)tr < (unsigned)lv.rows) & __builtin_amdgcn_ballot_w64((unsigned)tc < (unsigned)lv.cols);
        return;
    }
#endif
    X = fmaf(T.r02, pz, fmaf(T.r01, py, fmaf(T.r00, px, wc.tx)));
    Y = fmaf(T.r12, pz, fmaf(T.r11, py, fmaf(T.r10, px, wc.ty)));
    Z = fmaf(T.r22, pz, fmaf(T.r21, py, fmaf(T.r20, px, wc.tz)));
    rho2 = fmaf(Z, Z, Y * Y);
    d2 = fmaf(X, X, rho2);
    const float rho = sqrt_rn(rho2, inv_rho);       // inv_rho: float32 DATA for the Jacobian (consume_stage), not index work
    const float ax = fabsf(X), ay = fabsf(Y), az = fabsf(Z);
    const float mxp = fmaxf(fmaxf(ax, rho), 1e-9f), mnp = __builtin_amdgcn_fmed3f(ax, rho, 0.f);   // min of two non-negatives
    const float mxt = fmaxf(fmaxf(ay, az), 1e-9f), mnt = fminf(ay, az);
    const float r = rcp_rn(mxp * mxt);
    const float tp = mnp * (r * mxt);
    const float tt = mnt * (r * mxp);
    float phi_trg = atan_unit(tp);
    if (ax > rho) phi_trg = 1.57079637f - phi_trg;
    phi_trg = copysignf(phi_trg, X);
    const float theta = atan2_from_t(Y, Z, ay, az, tt);
    tr = round_index(fmaf(phi_trg, -lv.angle_res_inv, wc.half_nRows));
    tc = round_index(fmaf(theta, lv.angle_res_inv, wc.pi_k));
    // predicates of the per-pixel pass are kept as 64-bit lane masks (the compares' own SGPR results, combined on the scalar unit): the
    // ballot of a COMBINED bool costs a v_cndmask + v_cmp pair per use with this compiler
    vis_mask = __builtin_amdgcn_ballot_w64((unsigned)tr < (unsigned)lv.rows) & __builtin_amdgcn_ballot_w64((unsigned)tc < (unsigned)lv.cols);
}
__device__ __forceinline__ unsigned warp_pixel(const PoseRT& T, const WarpConsts& wc, float px, float py, float pz,
                                               const LevelDev& lv, float& X, float& Y, float& Z, float& rho2, float& d2,
                                               bool& vis) {
    int tr, tc;
    float inv_rho;
    unsigned long long vis_mask;
    warp_pixel_rc(T, wc, px, py, pz, lv, X, Y, Z, rho2, d2, tr, tc, vis_mask, inv_rho);
    vis = __builtin_amdgcn_inverse_ballot_w64(vis_mask);
    return __umul24(tr, lv.cols) + (unsigned)tc;
}

// ---------------------------------------------------------------------------------------------------------
// DPP move helper (quad_perm / row modes) used by the wave reduction below.
// ---------------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}
// ---------------------------------------------------------------------------------------------------------
// Wave reduction of 32 values at once ("halving" butterfly): every step pairs lanes and, instead of reducing each value
// in every lane, lets the two partners keep complementary halves of the value list, so the number of live registers
// halves per step: 32 -> 16 (v_permlane32_swap) -> 8 (v_permlane16_swap) -> 4 (quads {0,1}|{2,3}, DPP row_ror:8 with
// bank masks) -> 2 (quad parity, DPP row_shl/shr:4) -> full sums inside each quad (quad_perm).  64 VALU operations
// instead of 32 x 6 = 192.  On return lane (row r, quad b, any q) holds in out[j], j = 0,1, the wave total of value
//     j + 2*(b & 1) + 4*(b >> 1) + 8*(r & 1) + 16*(r >> 1).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float add_ror8_banks01(float dst, float src) {    // dst (quads 0,1) = src + src[lane+8 within row]
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3" : "+v"(dst) : "v"(src));
    return dst;
}
__device__ __forceinline__ float add_ror8_banks23(float dst, float src) {    // dst (quads 2,3) = src + src[lane-8 within row]
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(dst) : "v"(src));
    return dst;
}
__device__ __forceinline__ float add_shl4_even_quads(float dst, float src) { // dst (quads 0,2) = src + src[lane+4]
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5" : "+v"(dst) : "v"(src));
    return dst;
}
__device__ __forceinline__ float add_shr4_odd_quads(float dst, float src) {  // dst (quads 1,3) = src + src[lane-4]
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa" : "+v"(dst) : "v"(src));
    return dst;
}
// The swaps are written as inline asm: with ROCm 7.2's hipcc the __builtin_amdgcn_permlane{16,32}_swap builtins
// lose their second result when both results feed one add (the add reads the first register twice;
// tools/ubench/reduce_test2.hip reproduces it).  s_nop 1 covers the VALU-write -> permlane-read hazard.
__device__ __forceinline__ float swap32_add(float a, float b) {   // lanes 0-31: sum of a's halves, lanes 32-63: of b's
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
// v(lane) + v(lane ^ 32) in every lane (float64): the two halves of a wave exchanged by v_permlane32_swap on the value's two words
__device__ __forceinline__ double add_other_half_d(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v), lo2 = lo, hi2 = hi;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(lo2));      // lo: lanes 0-31 own | lanes 32-63 own-32 ... see swap32_add
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(hi), "+v"(hi2));
    return __hiloint2double(hi, lo) + __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ float swap16_add(float a, float b) {   // even rows: a's row pair, odd rows: b's row pair
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ void wave_reduce32(const float v[32], float out[2]) {
    float u[16], t[8], w[4], r[2];
#pragma unroll
    for (int j = 0; j < 16; ++j) u[j] = swap32_add(v[j], v[j + 16]);   // lanes 0-31 keep v[j], lanes 32-63 keep v[j+16]
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = swap16_add(u[j], u[j + 8]);     // even rows keep u[j], odd rows keep u[j+8]
#pragma unroll
    for (int j = 0; j < 4; ++j) {       // quads 0,1 keep t[j], quads 2,3 keep t[j+4]
        float d = 0.f;
        d = add_ror8_banks01(d, t[j]);
        d = add_ror8_banks23(d, t[j + 4]);
        w[j] = d;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {       // even quads keep w[j], odd quads keep w[j+2]
        float d = 0.f;
        d = add_shl4_even_quads(d, w[j]);
        d = add_shr4_odd_quads(d, w[j + 2]);
        r[j] = d;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {       // the four lanes of a quad hold partial sums of the same value
        float x = r[j];
        x += dpp_f<0xB1>(x);
        x += dpp_f<0x4E>(x);
        out[j] = x;
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_eval: one fused pass over the source pixels of a level at pose st->cand.
//   METHOD: 0 photo, 1 depth, 2 photo+depth.  HG: also accumulate the 21+6 normal-equation terms.
// Work split: block b owns the contiguous pixel span [cb*chunk, (cb+1)*chunk) (cb = XCD-aware remap of b so
// that neighbouring spans, which gather neighbouring target rows, share an XCD L2); its 1024 lanes sweep the
// span in coalesced 1024-pixel steps (16 B/lane source records), accumulate in float32 registers, then reduce
// wave (DPP) -> block (LDS) and store 32 float64 partials.  No atomics: the final sum order is fixed.
//
// The pixel body is branch-free: the reference's `continue`s (invalid point, not visible, non-salient) become
// predicates that zero the pixel's contribution, so every load is unconditional (the gather index of a skipped
// pixel is clamped to 0) and the compiler can issue the next source record and both gathers early instead of
// serialising four dependent memory round trips behind divergent branches.
// ---------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------
// Buffer-addressed loads for the fused pass: a 128-bit resource descriptor in SGPRs + a 32-bit byte offset per lane
// replaces 64-bit per-lane address arithmetic (v_mad_u64_u32 / v_lshl_add_u64), and the hardware range check (offsets
// past num_records return 0) replaces the index clamps: a lane past the end of its span, or a pixel warped outside
// the image, may issue its load with whatever offset it has -- the value is never used and the load cannot fault.
// ---------------------------------------------------------------------------------------------------------
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float3v __attribute__((ext_vector_type(3)));
typedef float float4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// (whole-vector casts: with ROCm 7.2's hipcc __builtin_bit_cast(float, v.y) of an ext-vector ELEMENT reads element 0)
__device__ __forceinline__ float4 buf_load_f4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const float4v w = (float4v)__builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return make_float4(w.x, w.y, w.z, w.w);
}
__device__ __forceinline__ F3 buf_load_f3(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const float3v w = (float3v)__builtin_amdgcn_raw_buffer_load_b96(r, (int)byte_off, 0, 0);
    F3 o = {w.x, w.y, w.z};
    return o;
}
struct EvalBufs {
    __amdgpu_buffer_rsrc_t src, trgP, trgD;
    unsigned row_bytes;      // cols * 12
};

// Source records of one step of a span: the descriptor is rebased on the step's first pixel (wave-uniform: scalar arithmetic), so the
// per-lane offset is the loop-invariant tid * 16 and no vector instruction goes into the address; records past the end of the buffer
// read as zero (num_records shrinks with the base, 0 once the step lies behind the buffer).
__device__ __forceinline__ float4 load_src_step(const float4* __restrict__ src0, const int n_px, const int first_px, const unsigned tid16) {
    const int left = max(n_px - first_px, 0);
    return buf_load_f4(make_rsrc(src0 + first_px, (unsigned)left * 16u), tid16);
}

// ---------------------------------------------------------------------------------------------------------
// The source stream of the per-pixel pass, three forms (template parameter SRC):
//   0  {x, y, z, Isrc} records, 16 B per pixel (LUT_xyz_sphere precomputed per level, like the reference)
//   1  depth and intensity from the level's float planes (4 + 4 B per pixel) + the angle tables: the point is re-formed per pixel
//   2  {depth, Isrc} records (8 B per pixel) + the angle tables (the lock-step engine, which keeps no level-0 planes)
// Forms 1 / 2 move 8 B per source pixel less through the memory system (20 instead of 28 B/px photo, 32 instead of 40
// photo + depth); the point they form is LUT_xyz_sphere's, operation for operation (src_rec_px), so sums and poses are bit-identical
// to form 0.  Measured (round 4, tools/ab_libs.py, one box): every HBM-fed regime gains -- 4096 x 2048 photo + depth 56.9 -> 47.0 us,
// the 2048 x 1024 pass rotating over 8 pairs 13.7 -> 11.8 us -- the Infinity-Cache-resident 2048 x 1024 launch 3 % (14.7 -> 14.3 us
// fused) and the small levels lose 0.3 us each (more loads and a row / column cursor in a latency-bound launch): the host picks form 1
// for levels of kRecomputeMinPx pixels and more, the engine (always HBM-fed) form 2 on its large levels.
// ---------------------------------------------------------------------------------------------------------
struct SrcRaw { float d, I, st, ct, sp, cp; };
struct SrcCursor { int r, c, dr, dc; };      // (row, column) of the lane's pixel in the next step to load, and the per-step advance
__device__ __forceinline__ void cursor_advance(SrcCursor& k, const int cols) {
    k.c += k.dc; k.r += k.dr;
    const bool wrap = k.c >= cols;
    k.c = wrap ? k.c - cols : k.c;
    k.r = wrap ? k.r + 1 : k.r;
}
template <int SRC> struct SrcForm;
template <> struct SrcForm<0> {
    typedef float4 T;
    static __device__ __forceinline__ void cursor_init(SrcCursor&, const LevelDev&, int, int) {}
    static __device__ __forceinline__ float4 load(const LevelDev&, const float4* __restrict__ src0, const int n_px, const int first_px, const unsigned tid16, SrcCursor&) {
        return load_src_step(src0, n_px, first_px, tid16);
    }
    static __device__ __forceinline__ float4 value(const float4& q, const LevelDev&) { return q; }
};
__device__ __forceinline__ void src_tabs(SrcRaw& q, const LevelDev& lv, SrcCursor& k) {
    const float2v t = (float2v)__builtin_amdgcn_raw_buffer_load_b64(make_rsrc(lv.tabT, (unsigned)lv.cols * 8u), k.c << 3, 0, 0);
    const float2v p = (float2v)__builtin_amdgcn_raw_buffer_load_b64(make_rsrc(lv.tabP, (unsigned)lv.rows * 8u), k.r << 3, 0, 0);
    q.st = t.x; q.ct = t.y; q.sp = p.x; q.cp = p.y;
    cursor_advance(k, lv.cols);
}
__device__ __forceinline__ float4 src_point(const SrcRaw& q, const LevelDev& lv) {       // src_rec_px's arithmetic, bit for bit
    float4 o;
    o.w = q.I;
    const bool ok = lv.min_depth < q.d && q.d < lv.max_depth;
    const float t = -q.d * q.cp;
    o.x = ok ? q.d * q.sp : kInvalidPoint;
    o.y = t * q.st;
    o.z = t * q.ct;
    return o;
}
template <> struct SrcForm<1> {
    typedef SrcRaw T;
    static __device__ __forceinline__ void cursor_init(SrcCursor& k, const LevelDev& lv, int first_lane_px, int threads) {
        divmod24(first_lane_px, lv.cols, k.r, k.c);
        divmod24(threads, lv.cols, k.dr, k.dc);
    }
    static __device__ __forceinline__ SrcRaw load(const LevelDev& lv, const float4* __restrict__, const int n_px, const int first_px, const unsigned tid16, SrcCursor& k) {
        const int left = max(n_px - first_px, 0);
        const unsigned tid4 = tid16 >> 2;
        SrcRaw q;
        q.d = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(lv.depth_src + first_px, (unsigned)left * 4u), (int)tid4, 0, 0));
        q.I = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(lv.gray_src + first_px, (unsigned)left * 4u), (int)tid4, 0, 0));
        src_tabs(q, lv, k);
        return q;
    }
    static __device__ __forceinline__ float4 value(const SrcRaw& q, const LevelDev& lv) { return src_point(q, lv); }
};
template <> struct SrcForm<2> {
    typedef SrcRaw T;
    static __device__ __forceinline__ void cursor_init(SrcCursor& k, const LevelDev& lv, int first_lane_px, int threads) { SrcForm<1>::cursor_init(k, lv, first_lane_px, threads); }
    static __device__ __forceinline__ SrcRaw load(const LevelDev& lv, const float4* __restrict__, const int n_px, const int first_px, const unsigned tid16, SrcCursor& k) {
        const int left = max(n_px - first_px, 0);
        SrcRaw q;
        const float2v v = (float2v)__builtin_amdgcn_raw_buffer_load_b64(make_rsrc(lv.src2 + first_px, (unsigned)left * 8u), (int)(tid16 >> 1), 0, 0);
        q.d = v.x; q.I = v.y;
        src_tabs(q, lv, k);
        return q;
    }
    static __device__ __forceinline__ float4 value(const SrcRaw& q, const LevelDev& lv) { return src_point(q, lv); }
};

struct EvalAcc {
    float acc[27];       // 21 upper-triangle terms of H, 6 of g
    float e2p, e2d;      // per-lane float32 partial sums of squared weighted residuals
    int   nP, nD, nVis;  // wave-uniform counts (scalar registers: popcount of the predicate ballots)
};

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
// (the wave-mask form of the ballot: HIP's __ballot goes through an integer compare of the predicate, which costs the per-pixel loop a
// v_cndmask + v_cmp pair per count; this one is the compare's own SGPR mask)
__device__ __forceinline__ unsigned long long ballot_mask(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int ballot_count(bool p) { return __builtin_popcountll(ballot_mask(p)); }

// weightHuber (RPI.h:545-554) with the hardware sqrt / rcp approximations (1 ulp): weights are float32 data,
// not index work.
__device__ __forceinline__ float weight_huber_fast(float error, float k) {
    const float ea = fabsf(error);
    const float q = fmaf(2 * k, ea, -(k * k));
    const float w = q * fast_rsq(q * ea * ea);       // sqrt(q) / |e| with one transcendental
    return ea < k ? 1.f : w;
}

// acc += [J | res]^T-products of one residual row J = (j3, p' x j3): jacobianT36 = [I | -skew(p')] makes the last
// three entries of every Jacobian row the cross product of p' with the first three (RPI.h:2994-2996, 3026).
__device__ __forceinline__ void accumulate_row(EvalAcc& A, float jx, float jy, float jz, float X, float Y, float Z,
                                               float res) {
#pragma clang fp contract(fast)
    float J[6];
    J[0] = jx; J[1] = jy; J[2] = jz;
    J[3] = Y * jz - Z * jy;
    J[4] = Z * jx - X * jz;
    J[5] = X * jy - Y * jx;
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b, ++k) A.acc[k] += J[a] * J[b];
#pragma unroll
    for (int a = 0; a < 6; ++a) A.acc[21 + a] += J[a] * res;
}

// The pixel body is split in two stages so that the loop can be software-pipelined:
//   warp_stage    source record -> warped point, target index, both gathers ISSUED
//   consume_stage gathered records -> residuals, Jacobian rows, normal-equation terms
// k_eval runs warp_stage of pixel i+1 before consume_stage of pixel i: the gather latency of one pixel hides behind
// the arithmetic of its neighbour instead of stalling the wave.
struct PixW {
    float X, Y, Z, rho2, d2, isrc, inv_rho;   // inv_rho: the reciprocal-square-root estimate sqrt_rn(rho2) started from
    unsigned long long vis;                    // "visible" of the wave's 64 pixels as a lane mask (an SGPR pair: no flag register, no select)
    F3    tp, td;
};

// CHECK = false: every lane's pixel lies inside the span (all steps but the last of a span): no in_range test.
template <int METHOD, bool CHECK = true, int SRC = 0>
__device__ __forceinline__ void warp_stage(const typename SrcForm<SRC>::T sraw, const bool in_range, const PoseRT& T, const WarpConsts& wc,
                                           const LevelDev& lv, const EvalBufs& bufs, PixW& w) {
    const float4 s = SrcForm<SRC>::value(sraw, lv);
    int tr, tc;
    warp_pixel_rc(T, wc, s.x, s.y, s.z, lv, w.X, w.Y, w.Z, w.rho2, w.d2, tr, tc, w.vis, w.inv_rho);
    w.vis &= __builtin_amdgcn_ballot_w64(s.x != kInvalidPoint);
    if (CHECK) w.vis &= __builtin_amdgcn_ballot_w64(in_range);
    w.isrc = s.w;
    // tie the copy of the source intensity to the end of the warp arithmetic: scheduled earlier it would sit in
    // front of the whole stage and wait for the youngest load (vmcnt(0)) instead of the one this stage needs
    asm volatile("" : "+v"(w.isrc), "+v"(tc));
    // unconditional gathers, issued as soon as the index is known; an invisible pixel's offset is arbitrary but
    // range-checked by the buffer descriptor
    const unsigned off = __umul24(tr, bufs.row_bytes) + __umul24(tc, 12u);      // byte offset of the 12-byte target record
    if (METHOD != 1) w.tp = buf_load_f3(bufs.trgP, off);
    if (METHOD != 0) w.td = buf_load_f3(bufs.trgD, off);
}

template <int METHOD, bool HG>
__device__ __forceinline__ void consume_stage(PixW& w, const LevelDev& lv, const EvalConsts& ec, EvalAcc& A) {
    // keep each gather one 12-byte load: without this the compiler splits it and sinks the intensity / depth
    // dword under the saliency branch, adding a dependent memory round trip per pixel
    if (METHOD != 1) asm volatile("" : "+v"(w.tp.a), "+v"(w.tp.b), "+v"(w.tp.c));
    if (METHOD != 0) asm volatile("" : "+v"(w.td.a), "+v"(w.td.b), "+v"(w.td.c));
    const float X = w.X, Y = w.Y, Z = w.Z;
    const float d2 = w.d2;
    A.nVis += __builtin_popcountll(w.vis);

    // rows of jacobianProj23 (RPI.h:3000-3016) in terms of rho^2 = Y^2+Z^2 and d^2 = |p'|^2 (k = angle_res_inv):
    //   d c'/d(y,z) = k (Z, -Y) / rho^2
    //   d r'/d(x,y,z) = k (-rho^2, X Y, X Z) / (rho d^2)
    // (algebraically what the reference writes with 1/z, 1/(1+y^2/z^2), 1/sqrt(1-x^2/d^2)); float32 data, so the
    // hardware reciprocal square roots are used.  A weighted gradient (gx, gy) times this 2x3 matrix is, with
    // u = gx k / rho^2 and v = gy k / (rho d^2):   ( -v rho^2,  v X Y + u Z,  v X Z - u Y ).
    // inv_rho = 1 / rho is the estimate the warp's sqrt_rn(rho^2) started from (the same v_rsq_f32 of the same operand: one
    // transcendental per pixel less than asking again).
    float k_rho2 = 0.f, k_d2r = 0.f;
    const float dist_inv = fast_rsq(d2);
    if (HG && METHOD != 0) {            // photo + depth: both rows share the two factors (photo alone folds them into its row, below)
#pragma clang fp contract(fast)
        const float inv_rho = w.inv_rho;
        k_rho2 = lv.angle_res_inv * (inv_rho * inv_rho);
        k_d2r = (lv.angle_res_inv * (dist_inv * dist_inv)) * inv_rho;
    }

    unsigned long long photo_skip = 0ull;   // `continue` at RPI.h:2690 / 3039 also skips the depth term of the pixel
    if (METHOD != 1) {
        const float tgx = w.tp.b, tgy = w.tp.c;
        const unsigned long long nonsal = __builtin_amdgcn_ballot_w64(fabsf(tgx) < ec.thr_photo) & __builtin_amdgcn_ballot_w64(fabsf(tgy) < ec.thr_photo);
        photo_skip = nonsal;
        const unsigned long long ok = w.vis & ~nonsal;
        A.nP += __builtin_popcountll(ok);
        if (__builtin_amdgcn_inverse_ballot_w64(ok)) {
#pragma clang fp contract(fast)
            const float photoDiff = w.tp.a - w.isrc;
            const float wpf = weight_huber_fast(photoDiff, ec.sigma_photo) * ec.sigma_photo_inv_f;
            const float res = wpf * photoDiff;
            A.e2p += res * res;
            if (HG) {
                // (w * grad) * jacobianProj23
                float u, v;
                if (METHOD == 0) {      // one row per pixel: w k / rho once, two multiplications fewer than through k_rho2 / k_d2r
                    const float bq = (wpf * lv.angle_res_inv) * w.inv_rho;
                    u = tgx * (bq * w.inv_rho);
                    v = tgy * (bq * (dist_inv * dist_inv));
                } else {
                    u = (wpf * tgx) * k_rho2;
                    v = (wpf * tgy) * k_d2r;
                }
                const float vX = v * X;
                accumulate_row(A, -v * w.rho2, vX * Y + u * Z, vX * Z - u * Y, X, Y, Z, res);
            }
        }
    }
    if (METHOD != 0) {
        const float depth2 = w.td.a;
        const float tdx = w.td.b, tdy = w.td.c;
        const unsigned long long nonsal = __builtin_amdgcn_ballot_w64(fabsf(tdx) < ec.thr_depth) & __builtin_amdgcn_ballot_w64(fabsf(tdy) < ec.thr_depth);
        const unsigned long long ok = w.vis & ~photo_skip & ~nonsal & __builtin_amdgcn_ballot_w64(fabsf(depth2) < INFINITY);      // isfinite
        A.nD += __builtin_popcountll(ok);
        if (__builtin_amdgcn_inverse_ballot_w64(ok)) {
#pragma clang fp contract(fast)
            float dist = d2 * dist_inv;                                 // |p'|: rsq estimate + one Newton step (< 1 ulp)
            dist = fmaf(0.5f * dist_inv, fmaf(-dist, dist, d2), dist);
            const float depthDiff = depth2 - dist;
            const float sd = ec.sigma_depth * depth2;
            const float wd = weight_huber_fast(depthDiff, sd) * fast_rcp(sd);
            const float res = wd * depthDiff;
            A.e2d += res * res;
            if (HG) {
                // wd * (dgrad * jacobianProj23 - p'/dist)   (RPI.h:3080-3083)
                const float u = (wd * tdx) * k_rho2, v = (wd * tdy) * k_d2r;
                const float vX = v * X, sdi = wd * dist_inv;
                accumulate_row(A, -v * w.rho2 - sdi * X, (vX * Y + u * Z) - sdi * Y, (vX * Z - u * Y) - sdi * Z, X, Y, Z, res);
            }
        }
    }
}

// Argument order: the scalars the first instructions need (state pointer for the gate / pose loads, source base and span for
// the first record loads) lead the list so that the kernarg preload (build flag -amdgpu-kernarg-preload-count) delivers
// them in SGPRs at wave start; the structs follow and are fetched while those first loads are in flight.
// eval_block is the body of one workgroup; k_eval (one pair per launch) and k_eval_b (blockIdx.y = slot of a lock-step batch of
// pairs, sequence_engine.h) call it with their own base pointers.
// THREADS = 1024 (one pair per launch: the Infinity-Cache-resident regime, one block per CU) or 512 (the lock-step batch, HBM-fed:
// two blocks of different pairs share a CU, the tail of one overlaps the body of the other -- 9.3 -> 10.0 k alignments/s on one
// engine).  A 512-thread block computes EXACTLY what the 1024-thread block does: thread t plays the lanes t and t + 512 of the
// 1024-lane layout with an accumulator set for each (the ping-pong stages of the loop are those two lanes' pixels), its wave w
// stands for the waves w and w + 8, and the block sum runs over the same 16 wave rows in the same order -- bit-identical sums.
#ifdef RGBD360_EVAL_STAMPS
#define ESTAMP(i) es[i] = __builtin_amdgcn_s_memrealtime() - es0
#else
#define ESTAMP(i)
#endif
// Everything behind the gate: the software-pipelined pixel loop and the block reduction.  On entry the first warp stage (wA, the
// pixel at i) has been issued and sB holds the source record of the pixel at i + THREADS.
template <int METHOD, bool HG, int THREADS, int SRC = 0>
__device__ __forceinline__ void eval_span(const PoseRT& T, const WarpConsts& wc, const LevelDev& lv, const EvalConsts& ec,
                                          const EvalBufs& bufs, const float4* __restrict__ src0, const int n_px, const int base, const int end,
                                          const int b, const int nb, PixW& wA, typename SrcForm<SRC>::T sB, SrcCursor cur, double* __restrict__ partials,
                                          unsigned long long* es, const unsigned long long es0) {
    constexpr int NS = kEvalThreads / THREADS;           // accumulator sets (lanes of the 1024-lane layout per thread)
    EvalAcc A[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
#pragma unroll
        for (int k = 0; k < 27; ++k) A[q].acc[k] = 0.f;
        A[q].e2p = A[q].e2d = 0.f;
        A[q].nP = A[q].nD = A[q].nVis = 0;
    }
    EvalAcc& AA = A[0];            // pixels of the even steps (stage A)
    EvalAcc& AB = A[NS - 1];       // pixels of the odd steps (stage B): the same set at 1024 threads, lane t + 512's at 512

    // Wave-uniform trip count (every lane stays active: the ballots count whole waves); lanes past the end of the
    // span process a clamped record with in_range = false.
    const int n_steps = NS * ((end - base + kEvalThreads - 1) / kEvalThreads);
    const int tid = (int)threadIdx.x;
    const unsigned tid16 = (unsigned)tid << 4;
    // WAVE REBALANCING (round 4).  A SIMD's issue arbiter favours its oldest wave: with one 64-pixel group per wave and step, the four
    // waves a SIMD holds leave the loop 6.0 / 6.4 / 7.0 / 8.0 us after its start (per-wave stamps, tools/eval_stamps.py: always in
    // this order), and the block's barrier -- the kernel -- waits for the youngest.  So the layout's four oldest waves (0-3: one per
    // SIMD) take over the LAST step's groups of its four youngest (12-15): 9 / 8 / 8 / 7 groups per wave at 2048 x 1024 instead of
    // 8 each.  In the 1024-lane layout: lanes 0-255 get an extra step whose pixels are those of lanes 768-1023 in the span's last
    // step (first pixel 256 short of a regular step's), lanes 768-1023 stop one step early.  A 512-thread block plays the same layout
    // (its waves 0-3 = layout waves 0-3 and 8-11, its waves 4-7 = 4-7 and 12-15; the extra half-step is an even one: set A, the
    // dropped one an odd one: set B), so the sums stay bit-identical between the two block shapes.  Spans of fewer than 4 steps keep
    // one group per wave and step.
    const int wave_id = tid >> 6;
#ifdef RGBD360_NO_REBAL
    const bool rebal = false;
#else
    const bool rebal = n_steps >= 4 * NS;
#endif
    const int my_steps = n_steps + (rebal ? ((wave_id < 4 ? 1 : 0) - (wave_id >= THREADS / 64 - 4 ? 1 : 0)) : 0);      // wave-uniform
    auto fpx = [&](int s) { return base + s * THREADS - ((rebal && s == n_steps) ? 256 : 0); };      // first pixel of (half-)step s
    auto load_step = [&](int s) {
        if (SRC != 0 && rebal && s == n_steps) SrcForm<SRC>::cursor_init(cur, lv, fpx(s) + tid, THREADS);      // the extra step breaks the cursor's stride
        return SrcForm<SRC>::load(lv, src0, n_px, fpx(s), tid16, cur);
    };
    // The loop is unrolled by two with ping-pong register
    // sets (wA / wB) so that no register copy forces an early wait: while the arithmetic of step k runs, the gathers
    // of step k+1 and the source record of step k+2 are in flight.
    PixW wB;
    typename SrcForm<SRC>::T sA = load_step(2);
    int k = 0;
#define RGBD360_EVAL_LOOP_BODY(CHECK)                                                                                    \
        warp_stage<METHOD, CHECK, SRC>(sB, tid < end - fpx(k + 1), T, wc, lv, bufs, wB);                                \
        sB = load_step(k + 3);                                                                                          \
        consume_stage<METHOD, HG>(wA, lv, ec, AA);                                                                      \
        warp_stage<METHOD, CHECK, SRC>(sA, tid < end - fpx(k + 2), T, wc, lv, bufs, wA);                                \
        sA = load_step(k + 4);                                                                                          \
        consume_stage<METHOD, HG>(wB, lv, ec, AB);
    // steady state: straight-line body (no control-flow joins, so the compiler's waits are counted, not vmcnt(0)).  Only the last NS
    // steps of a span (and the extra one) can be partial, so the steps this loop warps (k + 1, k + 2 <= n_steps - 1 - NS) need no span test.
    for (; k + 2 + NS < n_steps; k += 2) {
        RGBD360_EVAL_LOOP_BODY(false)
    }
    // the same body with the span test for the trips that touch the span's last steps
    for (; k + 2 < my_steps; k += 2) {
        RGBD360_EVAL_LOOP_BODY(true)
    }
#undef RGBD360_EVAL_LOOP_BODY
    // tail: one or two steps left, wA holds step k
    if (k + 1 < my_steps) {
        warp_stage<METHOD, true, SRC>(sB, tid < end - fpx(k + 1), T, wc, lv, bufs, wB);
        consume_stage<METHOD, HG>(wA, lv, ec, AA);
        consume_stage<METHOD, HG>(wB, lv, ec, AB);
    } else {
        consume_stage<METHOD, HG>(wA, lv, ec, AA);
    }
    // (Round 3, measured and dropped: FOUR register sets, unrolled by four -- gathers three steps ahead, source records six -- to
    // cover an HBM miss instead of an Infinity-Cache hit: 95 / 111 VGPRs, same sums, and SLOWER everywhere on one box, resident
    // 12.0 vs 11.7 us, HBM-fed 15.1 vs 13.8 us (photo), 17.5 vs 15.4 / 19.7 vs 17.6 us (photo + depth), 4096 x 2048 58.6 vs 57.1 us:
    // the loop is not waiting for a deeper queue, tools/ab_libs.py.)

    ESTAMP(2);
#ifdef RGBD360_EVAL_STAMPS
    if ((threadIdx.x & 63) == 0)      // every wave's own "loop done" (diagnostic rows nb + 32 ...: 16 values per block)
        partials[(size_t)(nb + 32) * kNumPartials + (size_t)b * 16 + (threadIdx.x >> 6)] = (double)(__builtin_amdgcn_s_memrealtime() - es0);
#endif
    // ---- reduction: lanes -> wave (halving butterfly, f32) -> block (f64 via LDS) -> one partial row ----
    __shared__ double red[kEvalThreads / 64][kNumPartials];
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        const int wave = (int)(threadIdx.x >> 6) + q * (THREADS / 64);      // the wave of the 1024-lane layout this set belongs to
        float v[32], out[2];
#pragma unroll
        for (int k = 0; k < 27; ++k) v[k] = HG ? A[q].acc[k] : 0.f;
        v[P_E2P] = A[q].e2p;
        v[P_E2D] = A[q].e2d;
        v[P_NP] = v[P_ND] = v[P_NVIS] = 0.f;        // the counts are wave-uniform scalars, stored below
        wave_reduce32(v, out);
        if ((lane & 3) == 0) {
            const int row = lane >> 4, quad = (lane >> 2) & 3;
            const int idx = 2 * (quad & 1) + 4 * (quad >> 1) + 8 * (row & 1) + 16 * (row >> 1);
            if (idx + 0 < P_NP) red[wave][idx + 0] = (double)out[0];
            if (idx + 1 < P_NP) red[wave][idx + 1] = (double)out[1];
        }
        if (lane == 63) {
            red[wave][P_NP] = (double)A[q].nP;
            red[wave][P_ND] = (double)A[q].nD;
            red[wave][P_NVIS] = (double)A[q].nVis;
        }
    }
    ESTAMP(3);
    __syncthreads();
    if (threadIdx.x < kNumPartials) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < kEvalThreads / 64; ++w) v += red[w][threadIdx.x];
        partials[(size_t)b * kNumPartials + threadIdx.x] = v;
    }
#ifdef RGBD360_EVAL_STAMPS
    ESTAMP(4);
    if (threadIdx.x == 0 && (b == 0 || b == nb - 1)) {      // diagnostic rows behind the partial table
        double* o = partials + (size_t)(nb + (b == 0 ? 0 : 1)) * kNumPartials;
        for (int k = 0; k < 5; ++k) o[k] = (double)es[k];
        o[5] = (double)es0;
        }
    if (threadIdx.x == 0) {     // per-block (start, end) of the last launch: rows nb+8 ...
        double* pb = partials + (size_t)(nb + 8) * kNumPartials + 2 * b;
        pb[0] = (double)es0;
        pb[1] = (double)(es0 + es[4]);
    }
    if (threadIdx.x == 0 && b == 0) {
        {       // history of (start, end) of block 0 over consecutive launches: rows nb+2 ...
            unsigned long long* cnt = reinterpret_cast<unsigned long long*>(partials + (size_t)(nb + 2) * kNumPartials);
            const unsigned long long slot = atomicAdd(cnt, 1ull) % 60;
            double* h = partials + (size_t)(nb + 3) * kNumPartials + 2 * slot;
            h[0] = (double)es0;
            h[1] = (double)(es0 + es[4]);
        }
    }
#endif
}

template <int METHOD, bool HG, int THREADS = kEvalThreads, int SRC = 0>
__device__ __forceinline__ void eval_block(const GNState* __restrict__ st, const float4* __restrict__ src0, const int n_px,
                                           const int chunk, const int level, const int nb_arg, double* __restrict__ partials,
                                           const LevelDev& lv, const EvalConsts& ec) {
    unsigned long long es[6] = {0, 0, 0, 0, 0, 0};
#ifdef RGBD360_EVAL_STAMPS
    const unsigned long long es0 = __builtin_amdgcn_s_memrealtime();
#else
    const unsigned long long es0 = 0;
#endif
    const int nb = nb_arg;
    const int b = blockIdx.x;
    const int cb = ((nb & 7) == 0) ? (b & 7) * (nb >> 3) + (b >> 3) : b;
    const int base = cb * chunk;
    const int end = min(base + chunk, n_px);
    EvalBufs bufs;
    bufs.src = make_rsrc(src0, (unsigned)n_px * 16u);
    bufs.trgP = make_rsrc(lv.trgP, (unsigned)lv.n * 12u);
    bufs.trgD = make_rsrc(lv.trgD, (unsigned)lv.n * 12u);
    bufs.row_bytes = (unsigned)lv.cols * 12u;
    const int i = base + (int)threadIdx.x;
    // the first two source records do not depend on the state: issue them before the scalar loads of done / pose
    static_assert(THREADS == 1024 || THREADS == 512, "eval_block: 1024-lane layout, played by 1024 or 512 threads");
    SrcCursor cur = {0, 0, 0, 0};
    SrcForm<SRC>::cursor_init(cur, lv, i, THREADS);
    const typename SrcForm<SRC>::T sA = SrcForm<SRC>::load(lv, src0, n_px, base, (unsigned)threadIdx.x << 4, cur);
    const typename SrcForm<SRC>::T sB = SrcForm<SRC>::load(lv, src0, n_px, base + THREADS, (unsigned)threadIdx.x << 4, cur);
    // gate and pose are fetched in ONE batch of scalar loads, in parallel with the two record loads above.  The gate is
    // only TESTED after the first warp stage: an early-exit branch up here makes the compiler sink every load behind it
    // (one dependent memory round trip per sunk batch, ~1 us each); the asm statement that ends warp_stage cannot be moved
    // across the branch, so this order survives.  A no-op launch (finished / other level) costs one warp stage.
    const int2 gate = *reinterpret_cast<const int2*>(&st->done);            // {done, level_active}
    const PoseRT T = load_pose(st->cand);
    const WarpConsts wc = make_warp_consts(T, lv);
    ESTAMP(0);
    PixW wA;
    warp_stage<METHOD, true, SRC>(sA, i < end, T, wc, lv, bufs, wA);
    asm volatile("" ::: "memory");                   // the loads issued so far stay above the gate
    if (gate.x | (gate.y != level)) return;          // speculatively enqueued launch of a finished / later level
    ESTAMP(1);
    eval_span<METHOD, HG, THREADS, SRC>(T, wc, lv, ec, bufs, src0, n_px, base, end, b, nb, wA, sB, cur, partials, es, es0);
}

template <int METHOD, bool HG, int SRC = 0>
__global__ __launch_bounds__(kEvalThreads) void k_eval(const GNState* __restrict__ st, const float4* __restrict__ src0, int n_px,
                                                        int chunk, int level, int nb_arg, double* __restrict__ partials,
                                                        LevelDev lv, EvalConsts ec) {
    eval_block<METHOD, HG, kEvalThreads, SRC>(st, src0, n_px, chunk, level, nb_arg, partials, lv, ec);
}

// Lock-step batch of pairs: blockIdx.y = slot.  Every per-slot buffer of a level is one slice of a single allocation, so the
// slot's pointers are arithmetic on kernel arguments (SGPRs at wave start through the kernarg preload) -- no descriptor table, no
// dependent load in front of the first record loads.  Work split, partial rows and summation order per slot are exactly those
// of k_eval, hence bit-identical sums.
constexpr int kEvalThreadsBatch = 512;
template <int METHOD, bool HG, int SRC = 0>
__global__ __launch_bounds__(kEvalThreadsBatch) void k_eval_b(const GNState* __restrict__ states, const float4* __restrict__ src0, int n_px,
                                                          int chunk, int level, int nb_arg, double* __restrict__ partials,
                                                          int partials_stride, LevelDev lv, EvalConsts ec) {
    const int slot = blockIdx.y;
    lv.trgP += (size_t)slot * (size_t)n_px;
    lv.trgD += (size_t)slot * (size_t)n_px;
    if (SRC == 2) lv.src2 += (size_t)slot * (size_t)n_px;
    eval_block<METHOD, HG, kEvalThreadsBatch, SRC>(states + slot, SRC == 0 ? src0 + (size_t)slot * (size_t)n_px : src0, n_px, chunk, level, nb_arg,
                                                   partials + (size_t)slot * (size_t)partials_stride, lv, ec);
}

// ---------------------------------------------------------------------------------------------------------
// k_level_init: entering a pyramid level (RPI.h:4590-4604): it = 0, update = (1,..,1), lambda = 1, the first
// pass is evaluated at the incoming pose.  use_pose != 0 loads `pose` (first level / stage calls).
// ---------------------------------------------------------------------------------------------------------
struct Pose16 {
    float v[16];
};
__device__ __forceinline__ void level_init_one(GNState* st, const Pose16& pose, int use_pose, int reset_all, int level);
struct FsInit {            // k_eval_fs: on != 0 = start a schedule at `pose` on the launch's level (what k_level_init(pose, reset_all) does)
    Pose16 pose;
    int on;
};
__global__ void k_level_init(GNState* st, Pose16 pose, int use_pose, int reset_all, int level) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    level_init_one(st, pose, use_pose, reset_all, level);
}
// Lock-step batch: block s initialises slot s; a slot outside live_mask (its span of pairs is exhausted) is parked -- done, and
// on a level no launch carries -- so that every later launch of the round is a no-op for it.
__global__ void k_level_init_b(GNState* states, Pose16 pose, int use_pose, int reset_all, int level, unsigned long long live_mask) {
    if (threadIdx.x != 0) return;
    GNState* st = states + blockIdx.x;
    if (!((live_mask >> blockIdx.x) & 1ull)) {
        st->done = 1;
        st->level_active = -1;
        st->status = 0;
        return;
    }
    level_init_one(st, pose, use_pose, reset_all, level);
}
__device__ __forceinline__ void level_init_one(GNState* st, const Pose16& pose, int use_pose, int reset_all, int level) {
    if (reset_all) {
        for (int k = 0; k < 36; ++k) st->H[k] = st->Hused[k] = 0.f;
        for (int k = 0; k < 6; ++k) st->g[k] = st->gused[k] = 0.f;
        for (int k = 0; k < 8; ++k) st->iters[k] = 0;
        st->status = 0;
        st->n_evals = 0;
        st->acc_e2p = st->acc_e2d = 0.0;
        st->acc_np = st->acc_nd = st->acc_nvis = st->used_nvis = st->used_npix = 0;
    } else {
        // entered speculatively right behind the previous level's launches: only proceed if that level really finished
        if (st->status != 0 || !st->done || st->level_active != level + 1) return;
    }
    st->level_active = level;
    st->pend_nb = 0;
    if (use_pose)
        for (int k = 0; k < 16; ++k) st->pose[k] = pose.v[k];
    for (int k = 0; k < 16; ++k) st->cand[k] = st->pose[k];
    for (int k = 0; k < 6; ++k) st->update[k] = 1.f;
    st->lambda = 1.0;
    st->it = 0;
    st->first = 1;
    st->done = 0;
    st->error = st->new_error = st->diff_error = 0.0;
}

// ---------------------------------------------------------------------------------------------------------
// Lane-parallel 6x6 kernels for k_solve: matrix columns live on lanes, rows in registers with compile-time
// indices (no scratch memory).  They perform the same float32 operations in the same order as gn::inverse6 /
// gn::rank6 (gn_math.h), which the tests compare against through k_gn_step and the CPU oracle.
// ---------------------------------------------------------------------------------------------------------
// Broadcast of one lane's value to the whole wave through a scalar register (v_readlane_b32): the lane index is
// wave-uniform everywhere below, so no LDS crossbar (ds_bpermute) round trip is needed.
__device__ __forceinline__ float bcast(float v, int lane_uniform) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane_uniform));
}

// Lanes 0-5 hold the columns of M, lanes 6-11 the columns of the identity; returns false when a pivot is zero.
// On return lanes 6-11 hold the columns of M^-1 in x[0..5].
__device__ __forceinline__ bool lu_inverse6_lanes(const float* M /*LDS, column-major*/, int lane, float x[6]) {
    float a[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) a[r] = lane < 6 ? M[lane * 6 + r] : ((lane - 6) == r ? 1.f : 0.f);
    bool ok = true;
    float rk[6];                    // reciprocals of the six pivots (wave-uniform): the elimination's multipliers AND the back substitution's divisors
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float ck[6];
#pragma unroll
        for (int r = k; r < 6; ++r) ck[r] = bcast(a[r], k);      // (rows above k are finished: round 5 stopped broadcasting them)
        // Pivot search on the SCALAR unit: the column's six values are wave-uniform (v_readlane results), and for finite values
        // |x| > |y| is the unsigned comparison of their bit patterns without the sign -- integer compares and selects the scalar ALU
        // does beside the vector pipe (as float compares they were 15 v_cmp + 30 v_cndmask of the solve's one busy wave).  Same pivot as
        // gn::inverse6's float comparison for every finite column (first of equal magnitudes wins in both).
        int piv = k;
        unsigned best = __float_as_uint(ck[k]) & 0x7fffffffu;
#pragma unroll
        for (int r = k + 1; r < 6; ++r) {
            const unsigned v = __float_as_uint(ck[r]) & 0x7fffffffu;
            if (v > best) {
                best = v;
                piv = r;
            }
        }
        if (best == 0u) ok = false;
        // the row swap under a SCALAR branch (piv is wave-uniform): plain register swaps in the one arm that runs -- as selects over every
        // candidate row they were 70 of the inverse's 285 vector instructions, each a link of the launch's longest dependent chain
        const int pv = __builtin_amdgcn_readfirstlane(piv);
        if (pv != k) {              // (normal equations: the diagonal entry is the pivot more often than not -- one scalar test then)
#pragma unroll
            for (int r = k + 1; r < 6; ++r)
                if (pv == r) {
                    asm volatile("v_swap_b32 %0, %1" : "+v"(a[k]), "+v"(a[r]));      // (an asm statement is not if-converted into selects)
                    const float t = ck[k]; ck[k] = ck[r]; ck[r] = t;
                }
        }
        // multipliers l = c_r / c_k through ONE exact reciprocal per step (the five IEEE division sequences were the longest
        // dependent chain of the kernel); l differs from the quotient by at most one rounding
        rk[k] = rcp_rn(ck[k]);
#pragma unroll
        for (int r = k + 1; r < 6; ++r) {
            const float l = ck[r] * rk[k];
            if (lane > k) a[r] -= l * a[k];
        }
    }
    // back substitution on the right-hand-side lanes: U[r][c] is row r of lane c; U[r][r] is step r's pivot (row r is final once step r
    // has run: later steps swap and update rows below it only), so 1 / U[r][r] is the reciprocal that step already formed
#pragma unroll
    for (int r = 5; r >= 0; --r) {
        float sacc = a[r];
#pragma unroll
        for (int c = r + 1; c < 6; ++c) sacc -= bcast(a[r], c) * x[c];
        x[r] = sacc * rk[r];
    }
    return ok;
}

// Rank of the 6x6 matrix whose columns sit on lanes 0-5 (column-pivoted Householder QR, Eigen thresholds).
__device__ __forceinline__ int qr_rank6_lanes(const float* M /*LDS, column-major*/, int lane) {
    const int col = lane < 6 ? lane : 5;
    float a[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) a[r] = M[col * 6 + r];
    float sq = 0.f;
#pragma unroll
    for (int r = 0; r < 6; ++r) sq += a[r] * a[r];
    float maxColSq = 0.f;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const float v = bcast(sq, c);
        maxColSq = v > maxColSq ? v : maxColSq;
    }
    const float threshold_helper = maxColSq * (gn::kEpsF * gn::kEpsF) / 6.f;
    float pivots[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float maxpivot = 0.f;
    int nonzero = 6;
    bool stopped = false;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float sk = 0.f;
#pragma unroll
        for (int r = k; r < 6; ++r) sk += a[r] * a[r];
        int best = k;
        float bestSq = -1.f;
#pragma unroll
        for (int c = k; c < 6; ++c) {
            const float v = bcast(sk, c);
            if (v > bestSq) {
                bestSq = v;
                best = c;
            }
        }
        if (!stopped && bestSq < threshold_helper * (float)(6 - k)) {
            nonzero = k;
            stopped = true;
        }
        if (!stopped) {
            // swap columns k and best (wave-uniform test: usually nothing to swap)
            if (best != k) {
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    const float from_best = bcast(a[r], best), from_k = bcast(a[r], k);
                    a[r] = (lane == k) ? from_best : ((lane == best) ? from_k : a[r]);
                }
            }
            float ck[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) ck[r] = bcast(a[r], k);
            float tailSq = 0.f;
#pragma unroll
            for (int r = k + 1; r < 6; ++r) tailSq += ck[r] * ck[r];
            const float c0 = ck[k];
            float beta, tau;
            float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (tailSq == 0.f) {
                tau = 0.f;
                beta = c0;
            } else {
                // exact sqrt / reciprocals in 6 / 3 instructions (sqrt_rn, rcp_rn); the quotients become products with an exact
                // reciprocal (one extra rounding each, far below the rank thresholds)
                beta = sqrt_rn(c0 * c0 + tailSq);
                if (c0 >= 0.f) beta = -beta;
                const float rden = rcp_rn(c0 - beta);
#pragma unroll
                for (int r = k + 1; r < 6; ++r) v[r] = ck[r] * rden;
                tau = (beta - c0) * rcp_rn(beta);
            }
            v[k] = 1.f;
            float dot = 0.f;
#pragma unroll
            for (int r = k; r < 6; ++r) dot += v[r] * a[r];
            dot *= tau;
            if (lane > k) {
#pragma unroll
                for (int r = k; r < 6; ++r) a[r] -= dot * v[r];
            }
            pivots[k] = beta;
            const float ab = fabsf(beta);
            maxpivot = ab > maxpivot ? ab : maxpivot;
        }
    }
    const float thr = maxpivot * (gn::kEpsF * 6.f);
    int rank = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) rank += (k < nonzero && fabsf(pivots[k]) > thr) ? 1 : 0;
    return rank;
}

// ---------------------------------------------------------------------------------------------------------
// k_solve: one block.  (1) fixed-order float64 reduction of the block partials; (2) the serial part of one
// loop trip of alignFrames360 (accept test, termination test, rank test, GN step, pose composition), with
// the rank test and the 6x6 inverse on two waves side by side.
// ---------------------------------------------------------------------------------------------------------
constexpr int kSolveThreads = 1024;
// LDS of one solve: the staged state, the per-thread-group partial sums and the hand-over slots of the three working waves.
struct SolveShared {
    GNState sst;
    double red[kSolveThreads / 64][kNumPartials];      // per wave: the sum of its two row groups (q = 2 w, 2 w + 1), see reduce_rows_to_lds
    float shH[36], shM[36], shInv[36], shE[16], shCand[16], shUpd[6];
    int shGo, shRank, shLuOk;
    unsigned long long stamp0, stamp[8];      // diagnostic build only (RGBD360_SOLVE_STAMPS)
};
#ifdef RGBD360_SOLVE_STAMPS
#define SOLVE_STAMP_T(i, t) if (threadIdx.x == (t)) sh.stamp[i] = __builtin_amdgcn_s_memrealtime() - sh.stamp0
#else
#define SOLVE_STAMP_T(i, t)
#endif
#define SOLVE_STAMP(i) SOLVE_STAMP_T(i, 0)

// The serial part of one loop trip of alignFrames360 on a state staged in LDS (sh.sst) whose partial rows have been summed per
// thread group into sh.red (a barrier behind both), in three steps so that the fused launch (k_eval_fs) can take the rank test
// off its critical path: solve_totals -> solve_waves -> (barrier) -> solve_finish.  solve_staged strings them together.
//
// solve_totals: column sums of sh.red into sst.tot (fixed order); returns the damping the rank test will use, read while nobody
// writes the staged state (wave 0 updates first / lambda in solve_waves).  A barrier behind it.
// A thread's sum s of its rows (thread = row group q = tid / 32, value v = tid % 32) -> the LDS table of solve_totals: the two row
// groups a wave holds (lanes v and v + 32) are added in registers first, so the serial chain of solve_totals is 16 adds, not 32.
__device__ __forceinline__ void reduce_rows_to_lds(SolveShared& sh, double s) {
    const double pair = add_other_half_d(s);
    if ((threadIdx.x & 63) < kNumPartials) sh.red[threadIdx.x >> 6][threadIdx.x & 31] = pair;
}
__device__ __forceinline__ float solve_totals(SolveShared& sh) {
    constexpr int Q = kSolveThreads / 64;
    const int tid = threadIdx.x;
    if (tid < kNumPartials) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < Q; ++k) t += sh.red[k][tid];
        sh.sst.tot[tid] = t;
    }
    const float lam_spec = (float)(sh.sst.first ? sh.sst.lambda : sh.sst.lambda / 5.0);
    __syncthreads();
    SOLVE_STAMP_T(1, 0);       // totals summed
    return lam_spec;
}

// (H + lambda diag H).rank() on the calling wave (all 64 lanes), RPI.h:4682.  Returns the rank (wave-uniform) and leaves it in sh.shRank
// for readers behind a barrier.
__device__ __forceinline__ int solve_rank_wave(SolveShared& sh, const float lam_spec) {
    const int lane = threadIdx.x & 63;
    const double* tot = sh.sst.tot;
    float hval = 0.f;
    if (lane < 36) {
        const int r = lane / 6, c = lane - 6 * r;
        const int aa = r < c ? r : c, bb = r < c ? c : r;
        hval = (float)tot[P_H + (aa * (13 - aa)) / 2 + (bb - aa)];
        sh.shM[lane] = (lane % 7 == 0) ? hval + lam_spec * hval : hval;      // H + lambda diag(H)
    }
    const int rk = __builtin_amdgcn_readfirstlane(qr_rank6_lanes(sh.shM, lane));
    if (lane == 0) sh.shRank = rk;
    return rk;
}

// solve_waves: three waves side by side -- wave 0 the bookkeeping, wave 1 the 6x6 inverse, the update, the SE(3) exponential and
// the candidate pose, wave 2 the rank test (rank_now; the fused launch runs it later, on one block only).  No barrier behind it.
__device__ __forceinline__ void solve_waves(SolveShared& sh, const SolveCfg& cfg, const float lam_spec, const bool rank_now) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    GNState* st = &sh.sst;
    const double* tot = sh.sst.tot;
    // ---- waves 1 and 2 start the 6x6 work speculatively, in parallel with the bookkeeping of wave 0: a step is only ever
    //      computed from the H of THIS pass (go implies take, see below), and the damping of the rank test is known from the
    //      state before the decision: lambda on the first pass of a level, lambda / 5 after an accepted step (a rejected
    //      step ends the level, so its result is simply dropped).  RPI.h:4682, 4693, 4718 ----
    if (cfg.mode == 0 && wave == 2 && rank_now) {
        solve_rank_wave(sh, lam_spec);
        SOLVE_STAMP_T(7, 128);     // rank done
    }
    if (cfg.mode == 0 && wave == 1) {
        float hval = 0.f;
        if (lane < 36) {
            const int r = lane / 6, c = lane - 6 * r;
            const int aa = r < c ? r : c, bb = r < c ? c : r;
            hval = (float)tot[P_H + (aa * (13 - aa)) / 2 + (bb - aa)];
        }
        {
            if (lane < 36) sh.shH[lane] = hval;
            float x[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const bool ok = lu_inverse6_lanes(sh.shH, lane, x);     // hessian.inverse()   RPI.h:4693
            if (lane >= 6 && lane < 12) {
#pragma unroll
                for (int r = 0; r < 6; ++r) sh.shInv[(lane - 6) * 6 + r] = x[r];
            }
            if (lane == 0) sh.shLuOk = ok ? 1 : 0;
            SOLVE_STAMP_T(5, 64);      // inverse done
            // update_pose = (-H^-1) * g, row `lane`, summed in column order; kept aside until wave 0 has decided (its test
            // of the loop condition still reads the previous update)
            float upd = 0.f;
            if (lane < 6) {
#pragma unroll
                for (int c = 0; c < 6; ++c) upd += (-sh.shInv[c * 6 + lane]) * (float)tot[P_G + c];
                sh.shUpd[lane] = upd;
            }
            // CPose3D::exp(update, true) (RPI.h:4697): every lane evaluates the two scalar coefficients, lane 4*j+i
            // assembles element (i,j) of the 4x4
            const double ux = (double)bcast(upd, 0), uy = (double)bcast(upd, 1), uz = (double)bcast(upd, 2);
            const double wx = (double)bcast(upd, 3), wy = (double)bcast(upd, 4), wz = (double)bcast(upd, 5);
            // sin(a)/a and (1 - cos a)/a^2 are series in a^2: no square root on the launch's critical path for the angles a Gauss-Newton
            // update has (round 5; a^2 < 0.25: the Maclaurin branch of gn::sinc_cosc, to which a^2 is handed directly -- it differs from
            // sqrt(a^2)^2 by an ulp of a double, invisible behind the cast to float32 of RPI.h:4697)
            const double theta_sq = wx * wx + wy * wy + wz * wz;
            double ca = 0.0, cb = 0.0;
            const bool rot = theta_sq >= (128 * 2.220446049250313e-16) * (128 * 2.220446049250313e-16);
            if (rot) {
                if (theta_sq < 0.25) gn::sinc_cosc_sq(theta_sq, ca, cb);
                else gn::sinc_cosc(sqrt(theta_sq), ca, cb);
            }
            if (lane < 16) {
                const int i = lane & 3, j = lane >> 2;
                double e = (i == j) ? 1.0 : 0.0;
                if (i < 3 && j < 3) {
                    if (rot) {
                        // W = skew(w), W2 = W W, element (i, j), written with selects: a W[i][k] lookup with a per-lane index puts
                        // the matrix into scratch memory (the only scratch use of the whole library).  Off the diagonal
                        // W2_ij = w_i w_j and W_ij = -/+ w_k (k the third index); on it W2_ii = -(w_a^2) - (w_b^2), added in the
                        // order of the matrix product.
                        const double wi = i == 0 ? wx : (i == 1 ? wy : wz), wj = j == 0 ? wx : (j == 1 ? wy : wz);
                        const int kk = 3 - i - j;
                        const double wk = kk == 0 ? wx : (kk == 1 ? wy : wz);
                        double Wij, W2ij;
                        if (i == j) {
                            Wij = 0.0;
                            const double first = i == 2 ? wy : wz, second = i == 0 ? wy : wx;      // k ascending, k != i
                            W2ij = (0.0 - first * first) - second * second;
                        } else {
                            Wij = ((j - i + 3) % 3 == 1) ? -wk : wk;
                            W2ij = wi * wj;
                        }
                        e += ca * Wij + cb * W2ij;
                    }
                } else if (j == 3 && i < 3) {
                    e = i == 0 ? ux : (i == 1 ? uy : uz);
                }
                sh.shE[lane] = (float)e;
            }
            // pose_estim_temp = exp(...).cast<float>() * pose_estim: whenever a step is taken the pose it starts from is the
            // candidate of this pass (accepted: pose := cand; first pass of a level: cand == pose)
            if (lane < 16) {
                const int c = lane >> 2, r = lane & 3;
                const float* P = sh.sst.cand;
                sh.shCand[lane] = ((sh.shE[0 * 4 + r] * P[c * 4 + 0] + sh.shE[1 * 4 + r] * P[c * 4 + 1]) + sh.shE[2 * 4 + r] * P[c * 4 + 2]) +
                               sh.shE[3 * 4 + r] * P[c * 4 + 3];
            }
            SOLVE_STAMP_T(6, 64);      // update, exponential, candidate pose done
        }
    }
    // ---- bookkeeping on wave 0: lane 0 takes the scalar decisions, lanes 0-41 move the 36 + 6 matrix entries ----
    if (wave == 0) {
        // normal equations at the evaluated pose (float like the reference's `hessian` / `gradient`): lane l < 36 owns
        // H(r,c), r = l/6, c = l%6 (upper-triangle slot a*(13-a)/2 + (b-a)); lanes 36-41 own g
        float hval = 0.f;
        if (lane < 36) {
            const int r = lane / 6, c = lane - 6 * r;
            const int aa = r < c ? r : c, bb = r < c ? c : r;
            hval = (float)tot[P_H + (aa * (13 - aa)) / 2 + (bb - aa)];
        } else if (lane < 42) {
            hval = (float)tot[P_G + lane - 36];
        }
        int take = 0, go = 0;
        if (lane == 0) {
            st->n_evals += 1;
            const double err2 = tot[P_E2P] + tot[P_E2D];
            const double nvalid = tot[P_NP] + tot[P_ND];
            const double new_error = cfg.occ == 0 ? sqrt(err2 / nvalid)      // RPI.h:2738
                                                  : sqrt(tot[P_E2P] / tot[P_NP]) + sqrt(tot[P_E2D] / tot[P_ND]);
            st->new_error = new_error;
            if (cfg.mode == 1) {
                take = 1;
            } else {
                bool stop = false;
                if (st->first) {
                    st->first = 0;
                    if (nvalid == 0.0 || new_error != new_error) {      // no residuals (occlusion modes: 0/0 of an unused modality)
                        st->status = 2;
                        st->done = 1;
                        stop = true;
                    } else {
                        st->error = new_error;           // RPI.h:4599
                        st->diff_error = new_error;      // RPI.h:4605
                        take = 1;                        // cand == pose
                    }
                } else {
                    const double diff = st->error - new_error;   // RPI.h:4713
                    st->diff_error = diff;
                    if (cfg.forced || diff > cfg.tol_residual) {  // RPI.h:4715-4722
                        st->lambda = st->lambda / 5.0;
                        st->error = new_error;
                        st->it += 1;
                        st->iters[cfg.level & 7] = st->it;
                        take = 2;                        // 2: also promote cand -> pose
                    }
                }
                if (take) {
                    st->acc_e2p = tot[P_E2P];
                    st->acc_e2d = tot[P_E2D];
                    st->acc_np = (long long)tot[P_NP];
                    st->acc_nd = (long long)tot[P_ND];
                    st->acc_nvis = (long long)tot[P_NVIS];
                }
                if (!stop) {
                    // while(it < maxIters && update_pose.norm() > tol_update && diff_error > tol_residual)   RPI.h:4611
                    float un = 0.f;
                    for (int i = 0; i < 6; ++i) un += st->update[i] * st->update[i];
                    un = sqrtf(un);
                    go = cfg.forced || (st->it < cfg.max_iters && (double)un > cfg.tol_update && st->diff_error > cfg.tol_residual);
                    if (!go) st->done = 1;
                }
                if (go) {
                    st->used_nvis = st->acc_nvis;
                    st->used_npix = cfg.n_pixels;
                }
            }
            sh.shGo = go;
        }
        take = __builtin_amdgcn_readfirstlane(take);
        go = __builtin_amdgcn_readfirstlane(go);
        if (take == 2 && lane < 16) st->pose[lane] = st->cand[lane];
        if (take) {
            if (lane < 36) st->H[lane] = hval;
            else if (lane < 42) st->g[lane - 36] = hval;
        }
        if (go) {
            // a step is only ever computed right after its pose was taken, so hval is H / g at `pose`;
            // record them as "used" (what the reference's `hessian` / `SSO` members hold afterwards)
            if (lane < 36) {
                st->Hused[lane] = hval;
            } else if (lane < 42) {
                st->gused[lane - 36] = hval;
            }
        }
    }
    SOLVE_STAMP(2);
}

// solve_finish (behind a barrier that follows solve_waves): commits the step or the ILL-POSED status, hands a finished level over to
// the next finer one.  (k_eval_fs restates these few decisions without barriers: keep the two in step.)
__device__ __forceinline__ void solve_finish(SolveShared& sh, const SolveCfg& cfg) {
    const int tid = threadIdx.x;
    GNState* st = &sh.sst;
    if (sh.shGo) {
        const bool ill = (sh.shRank != 6) || !sh.shLuOk;
        if (ill) {
            if (tid == 0) {
                st->status = 1;      // "The problem is ILL-POSED": relPose = pose_estim, return   RPI.h:4684-4689
                st->done = 1;
            }
        } else if (tid < 16) {
            sh.sst.cand[tid] = sh.shCand[tid];
            if (tid < 6) sh.sst.update[tid] = sh.shUpd[tid];
        }
        SOLVE_STAMP(3);
        __syncthreads();
    }
    // A level that has just finished hands over to the next finer one right here (RPI.h:4590-4604: it = 0, update = (1,..,1),
    // lambda = 1, first pass at the pose reached so far): the next launch in the queue, tagged with that level, finds it
    // active -- no separate initialisation launch between levels.
    if (cfg.mode == 0 && !cfg.forced && cfg.level > 0 && sh.sst.done && sh.sst.status == 0) {      // uniform
        __syncthreads();
        if (tid < 16) sh.sst.cand[tid] = sh.sst.pose[tid];
        if (tid < 6) sh.sst.update[tid] = 1.f;
        if (tid == 0) {
            sh.sst.level_active = cfg.level - 1;
            sh.sst.lambda = 1.0;
            sh.sst.it = 0;
            sh.sst.first = 1;
            sh.sst.done = 0;
            sh.sst.error = sh.sst.new_error = sh.sst.diff_error = 0.0;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void solve_staged(SolveShared& sh, const SolveCfg& cfg) {
    const float lam_spec = solve_totals(sh);
    solve_waves(sh, cfg, lam_spec, true);
    __syncthreads();            // bookkeeping (wave 0), inverse (wave 1) and rank (wave 2) are all done
    solve_finish(sh, cfg);
}

__device__ __forceinline__ void solve_block(GNState* st_g, const double* __restrict__ partials, const int nb, const SolveCfg& cfg) {
    // The state is staged through LDS: one coalesced read while the partials are being reduced, one coalesced
    // write-back at the end; the single-lane bookkeeping then never waits on global memory.
    __shared__ SolveShared sh;
    constexpr int kStateWords = sizeof(GNState) / 4;
    static_assert(sizeof(GNState) % 4 == 0 && kStateWords <= kSolveThreads, "GNState staging");
    const int tid = threadIdx.x;
#ifdef RGBD360_SOLVE_STAMPS
    if (tid == 0) sh.stamp0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (tid < kStateWords) reinterpret_cast<int*>(&sh.sst)[tid] = reinterpret_cast<const int*>(st_g)[tid];
    const int v = tid % kNumPartials, q = tid / kNumPartials;
    constexpr int Q = kSolveThreads / kNumPartials;
    // rows q, q+Q, q+2Q, ... of the partial table: all loads of a batch are issued before the first add (one
    // memory round trip per 16 rows instead of one per row); the summation order stays fixed.
    double s = 0.0;
    for (int b0 = q; b0 < nb; b0 += 16 * Q) {
        double tmp[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int b = b0 + j * Q;
            tmp[j] = b < nb ? partials[(size_t)b * kNumPartials + v] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) s += tmp[j];
    }
    reduce_rows_to_lds(sh, s);
    __syncthreads();
    SOLVE_STAMP(0);
    if (cfg.mode == 0 && (sh.sst.done || sh.sst.level_active != cfg.level)) {            // uniform: finished / other level
        if (cfg.host_state) {                               // the schedule's last launch reports even when it has nothing to do
            if (tid < 64) {
                for (int w = tid; w < kStateWords; w += 64) reinterpret_cast<int*>(cfg.host_state)[w] = reinterpret_cast<const int*>(&sh.sst)[w];
                __threadfence_system();
            }
            __syncthreads();
            if (tid == 0) __hip_atomic_store(cfg.host_tag, cfg.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    solve_staged(sh, cfg);
#ifdef RGBD360_SOLVE_STAMPS
    if (tid == 0) {
        sh.stamp[4] = __builtin_amdgcn_s_memrealtime() - sh.stamp0;
        for (int i = 0; i < 8; ++i) sh.sst.stamps[i] = sh.stamp[i];
    }
    __syncthreads();
#endif
    if (tid < kStateWords) reinterpret_cast<int*>(st_g)[tid] = reinterpret_cast<const int*>(&sh.sst)[tid];
    if (cfg.host_state) {                                   // uniform
        // (one wave writes the state's ~230 words and runs the one system-scope fence: sixteen waves each asking for a write-back cost
        // the launch 3-4 us)
        if (tid < 64) {
            for (int w = tid; w < kStateWords; w += 64) reinterpret_cast<int*>(cfg.host_state)[w] = reinterpret_cast<const int*>(&sh.sst)[w];
            __threadfence_system();
        }
        __syncthreads();
        if (tid == 0) __hip_atomic_store(cfg.host_tag, cfg.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ __launch_bounds__(kSolveThreads) void k_solve(GNState* st_g, const double* __restrict__ partials, int nb,
                                                          SolveCfg cfg) {
    solve_block(st_g, partials, nb, cfg);
}
// Lock-step batch: one block per slot, each the k_solve of its own pair (same reduction order, same arithmetic).
__global__ __launch_bounds__(kSolveThreads) void k_solve_b(GNState* states, const double* __restrict__ partials, int partials_stride,
                                                            int nb, SolveCfg cfg) {
    solve_block(states + blockIdx.x, partials + (size_t)blockIdx.x * (size_t)partials_stride, nb, cfg);
}

// ---------------------------------------------------------------------------------------------------------
// Fused-solve schedule of the single-pair path (one launch per Gauss-Newton iteration instead of two).
// The solve of pass k is the PROLOGUE of the launch that runs pass k+1: every block of that launch reduces the partial rows of
// the previous pass and takes the loop trip of alignFrames360 (RPI.h:4611-4722) itself -- redundantly, in the same fixed order,
// so all blocks arrive at the same state bit for bit -- and then evaluates its span at the candidate pose it has just computed.
// One pair keeps the 256 CUs busy for ~9 us per pass and leaves them idle during the ~5 us single-block k_solve launch and
// both launch boundaries around it; here the serial part runs where the next pass starts, after ONE boundary, while the
// pose-independent source records of the pass are already in flight.  No cross-block communication inside a launch: the
// kernel boundary publishes the partial rows, state and partial table are double-buffered (a launch reads buffers A, block 0
// writes the new state to B, all blocks write their rows to B) so that no block reads what another block of the same launch
// writes.  What a pass leaves behind is described by the state itself (pend_nb rows, pend_npix pixels, on level
// level_active, at pose cand).  The arithmetic is solve_staged / eval_span, shared with k_solve / k_eval: poses are
// bit-identical to the two-launch schedule.  The lock-step sequence engine keeps k_solve_b: there the CUs are never idle and
// one solve launch already serves all pairs in flight.
// ---------------------------------------------------------------------------------------------------------
// Stages the state and sums the pending pass's partial rows.  Everything is requested at once, in the first instructions of the
// launch: the first kPendingRows rows of the table (always that many: the row count of the pending pass is only known with the state,
// and waiting for it would put two dependent round trips in front of the solve; rows >= pend_nb are dropped from the sum, the table
// is allocated with at least kPendingRows rows), the state words, and the pending row count itself through the scalar unit.
// Straight-line code: a launch's first instruction-cache lines arrive cold (0.7 us passed before the loads were out when each
// load sat behind its own test).  Returns pend_nb; a barrier behind the LDS writes.
constexpr int kPendingRows = 256;                 // rows loaded per launch = the grid cap of the single-pair pass
constexpr int kMaxPendingRows = kPendingRows;     // the fused schedule needs every level's block count <= this
// rows_hint (a kernel argument: in an SGPR when the launch starts) = an upper bound of the pending row count the HOST knows when it
// enqueues the launch -- the block count of the launch in front of it.  The launches of the small pyramid levels leave 8 - 64 rows:
// with the bound they request one batch of 32 rows (8 KB per workgroup) instead of all 256 (64 KB through one CU's 64 B / clock L1 fill
// path: 0.5 us of the prologue of a launch that has 1 us of pixel work).  Rows beyond pend_nb add nothing either way: same sums.
__device__ __forceinline__ int stage_pending(SolveShared& sh, const GNState* __restrict__ st_in, const double* __restrict__ partials, const int rows_hint) {
    constexpr int kStateWords = sizeof(GNState) / 4;
    constexpr int Q = kSolveThreads / kNumPartials;
    constexpr int J = kPendingRows / Q;
    static_assert(J * Q == kPendingRows && J <= 16, "whole batches");
    const int tid = threadIdx.x;
    const int v = tid % kNumPartials, q = tid / kNumPartials;
    double tmp[J];
    tmp[0] = partials[(size_t)q * kNumPartials + v];
#pragma unroll
    for (int j = 1; j < J; ++j) tmp[j] = 0.0;
    // (fall-through = the full table: the level-0 launch runs straight-line code -- as an if / else with the short form first its loads
    // left 0.2 us later, behind a taken branch into a cold instruction-cache line)
    if (rows_hint > Q) {                        // uniform: a scalar compare on a preloaded argument
#pragma unroll
        for (int j = 1; j < J; ++j) tmp[j] = partials[(size_t)(q + j * Q) * kNumPartials + v];
    }
    int word = 0;
    if (tid < kStateWords) word = reinterpret_cast<const int*>(st_in)[tid];
    const int nb = st_in->pend_nb;        // uniform (scalar load)
    // The host's bound is checked, not trusted: a state that holds more pending rows than the short form requested gets the rest now
    // (a second round trip, never taken by the library's own schedules; without it those rows would silently count as zero).
    if (nb > Q && rows_hint <= Q) {             // uniform
#pragma unroll
        for (int j = 1; j < J; ++j) tmp[j] = partials[(size_t)(q + j * Q) * kNumPartials + v];
    }
#ifdef RGBD360_SOLVE_STAMPS
    if (tid == 0) sh.stamp[3] = __builtin_amdgcn_s_memrealtime();      // loads issued (absolute; made relative below)
#endif
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) s += (q + j * Q < nb) ? tmp[j] : 0.0;      // the order of solve_block's sum (its zero rows add nothing)
    reduce_rows_to_lds(sh, s);
    if (tid < kStateWords) reinterpret_cast<int*>(&sh.sst)[tid] = word;
#ifdef RGBD360_SOLVE_STAMPS
    if (tid == 0) {
        sh.stamp[5] = __builtin_amdgcn_s_memrealtime() - sh.stamp0;      // this wave's loads are back, sums in LDS (slot re-used: wave 1 overwrites it later)
        sh.stamp[3] -= sh.stamp0;
    }
#endif
    __syncthreads();
    return nb;
}

// wave-uniform pose out of LDS into scalar registers (the pixel loop reads it as SGPR operands, like the scalar loads of k_eval)
__device__ __forceinline__ float uniform_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// The solve half of a fused launch, behind stage_pending (and the schedule's initialisation): solves the pending pass on the block's
// own LDS copy of the state, decides what the new state is, hands every wave the pose the new state's `cand` holds (scalar registers)
// and lets block 0 write the new state (pend_nb = nb if the launch's own pass runs).  Returns whether the launch's level is the active,
// unfinished one.  Shared by k_eval_fs and the occlusion-aware schedule's k_occ_build_fs (occlusion_kernels.h).
// In two halves: fs_solve_decide -- the solve and what it means, by every wave -- and fs_write_state -- the rank verdict and the new
// state, by ONE wave (nothing the first half left in LDS may have been overwritten in between).
struct FsDecision {
    bool run, commit, ill, handover;
    int level_now;
    float lam_spec;
};
__device__ __forceinline__ FsDecision fs_solve_decide(SolveShared& sh, const int pend, const SolveCfg& cfg, const int level, PoseRT& T) {
    float lam_spec = 0.f;
    int solved_level = 0;
    if (pend > 0) {                             // uniform: the previous launch ran a pass
        SolveCfg c = cfg;
        c.level = solved_level = sh.sst.level_active;      // the level that pass belongs to (it ran because this level was active)
        c.n_pixels = sh.sst.pend_npix;
        lam_spec = solve_totals(sh);
        solve_waves(sh, c, lam_spec, /*rank_now=*/false);
        __syncthreads();
    }
    // What solve_finish would now do to the state -- commit the step (on the inverse's word: the rank test, the longest of the three
    // chains and almost never the one that says no, comes later), or ILL-POSED, or hand a finished level over to the next finer
    // one -- is DECIDED here by every wave from what the three working waves left in LDS, without another barrier; nothing in LDS
    // is written any more.  Only block 0 needs the new state itself: its wave 2 assembles it word by word on the way to memory.
    FsDecision d;
    const bool go = pend > 0 && sh.shGo != 0;
    d.commit = go && sh.shLuOk != 0;
    d.ill = go && !d.commit;
    d.handover = pend > 0 && !go && !cfg.forced && solved_level > 0 && sh.sst.done && sh.sst.status == 0;      // solve_finish's test
    d.level_now = d.handover ? solved_level - 1 : sh.sst.level_active;
    const bool done_now = d.handover ? false : (d.ill || sh.sst.done != 0);
    d.run = !done_now && d.level_now == level;      // uniform; k_eval's gate on the new state
    d.lam_spec = lam_spec;
    const float* P = d.commit ? sh.shCand : (d.handover ? sh.sst.pose : sh.sst.cand);      // the pose the new state's `cand` holds
    T.r00 = uniform_f(P[0]); T.r10 = uniform_f(P[1]); T.r20 = uniform_f(P[2]);
    T.r01 = uniform_f(P[4]); T.r11 = uniform_f(P[5]); T.r21 = uniform_f(P[6]);
    T.r02 = uniform_f(P[8]); T.r12 = uniform_f(P[9]); T.r22 = uniform_f(P[10]);
    T.tx = uniform_f(P[12]); T.ty = uniform_f(P[13]); T.tz = uniform_f(P[14]);
    return d;
}
// one wave (all 64 lanes): the rank verdict on the step just committed, then the new state.  ILL-POSED (RPI.h:4684-4689): status 1,
// level done, candidate and update as they were -- and no pending pass: the rows this launch writes are never read.
__device__ __forceinline__ void fs_write_state(SolveShared& sh, const FsDecision& d, const int nb, const int n_level_px, GNState* st_out) {
    constexpr int kStateWords = sizeof(GNState) / 4;
    const bool commit = d.commit, ill = d.ill, handover = d.handover, run = d.run;
    const int level_now = d.level_now;
    const int lane = threadIdx.x & 63;
    bool ill_late = false;
    if (commit) ill_late = solve_rank_wave(sh, d.lam_spec) != 6;      // (the return value: a lane that did not store shRank may not see it without a barrier)
    const bool commit_f = commit && !ill_late, ill_f = ill || ill_late, run_f = run && !ill_late;
    constexpr int kCand = offsetof(GNState, cand) / 4, kUpd = offsetof(GNState, update) / 4;
    constexpr int kErr = offsetof(GNState, error) / 4, kLam = offsetof(GNState, lambda) / 4;
    static_assert(offsetof(GNState, new_error) == offsetof(GNState, error) + 8 && offsetof(GNState, diff_error) == offsetof(GNState, error) + 16, "error block");
    for (int w = lane; w < kStateWords; w += 64) {      // one store per word
        int val = reinterpret_cast<const int*>(&sh.sst)[w];
        if (w >= kCand && w < kCand + 16) {
            if (commit_f) val = __builtin_bit_cast(int, sh.shCand[w - kCand]);
            if (handover) val = __builtin_bit_cast(int, sh.sst.pose[w - kCand]);
        }
        if (w >= kUpd && w < kUpd + 6) {
            if (commit_f) val = __builtin_bit_cast(int, sh.shUpd[w - kUpd]);
            if (handover) val = __builtin_bit_cast(int, 1.f);
        }
        if (ill_f && (w == offsetof(GNState, status) / 4 || w == offsetof(GNState, done) / 4)) val = 1;
        if (handover) {       // RPI.h:4590-4604: it = 0, update = (1,..,1), lambda = 1, first pass at the pose reached
            if (w == offsetof(GNState, level_active) / 4) val = level_now;
            if (w == offsetof(GNState, it) / 4 || w == offsetof(GNState, done) / 4) val = 0;
            if (w == offsetof(GNState, first) / 4) val = 1;
            if (w >= kErr && w < kErr + 6) val = 0;                  // error = new_error = diff_error = 0.0
            if (w == kLam) val = 0;                                   // lambda = 1.0 (little endian: low word, high word)
            if (w == kLam + 1) val = 0x3FF00000;
        }
        if (w == offsetof(GNState, pend_nb) / 4) val = run_f ? nb : 0;
        if (w == offsetof(GNState, pend_npix) / 4) val = n_level_px;
#ifdef RGBD360_SOLVE_STAMPS
        if (w >= (int)offsetof(GNState, stamps) / 4 && w < (int)offsetof(GNState, stamps) / 4 + 16) {
            const int si = (w - (int)offsetof(GNState, stamps) / 4) >> 1;      // slot 4 ("end"): now
            const unsigned long long sv = si == 4 ? __builtin_amdgcn_s_memrealtime() - sh.stamp0 : sh.stamp[si];
            val = (int)((w & 1) ? (sv >> 32) : (sv & 0xffffffffull));
        }
#endif
        reinterpret_cast<int*>(st_out)[w] = val;
    }
}
__device__ __forceinline__ bool fs_solve_and_publish(SolveShared& sh, const int pend, const SolveCfg& cfg, const int level, const int nb,
                                                     const int n_level_px, GNState* __restrict__ st_out, PoseRT& T) {
    const FsDecision d = fs_solve_decide(sh, pend, cfg, level, T);
    if (blockIdx.x == 0 && (threadIdx.x >> 6) == 2) fs_write_state(sh, d, nb, n_level_px, st_out);      // block 0, wave 2
    return d.run;
}

template <int METHOD, int SRC = 0>
__global__ __launch_bounds__(kEvalThreads) void k_eval_fs(const GNState* __restrict__ st_in, GNState* __restrict__ st_out,
                                                           const double* __restrict__ partials_in, double* __restrict__ partials_out,
                                                           const float4* __restrict__ src0, int n_px, int chunk, int level, int nb_arg,
                                                           int pend_rows_hint, LevelDev lv, EvalConsts ec, SolveCfg cfg, FsInit init) {
    static_assert(kEvalThreads == kSolveThreads, "the fused launch runs the solve on the pass's block");
    unsigned long long es[6] = {0, 0, 0, 0, 0, 0};
#ifdef RGBD360_EVAL_STAMPS
    const unsigned long long es0 = __builtin_amdgcn_s_memrealtime();
#else
    const unsigned long long es0 = 0;
#endif
    __shared__ SolveShared sh;
    constexpr int kStateWords = sizeof(GNState) / 4;
#ifdef RGBD360_SOLVE_STAMPS
    if (threadIdx.x == 0) sh.stamp0 = __builtin_amdgcn_s_memrealtime();
#endif
    // what the solve waits for is requested first (a wave's vector-memory operations complete in issue order) ...
    int pend = stage_pending(sh, st_in, partials_in, pend_rows_hint);
    if (init.on) {                              // uniform: the first launch of a schedule is its k_level_init too (every block initialises
        if (threadIdx.x == 0) level_init_one(&sh.sst, init.pose, 1, 1, level);      // its own LDS copy of the state, block 0 writes it out)
        __syncthreads();
        pend = 0;
    }
    SOLVE_STAMP(0);
    const int nb = nb_arg;
    const int b = blockIdx.x;
    const int cb = ((nb & 7) == 0) ? (b & 7) * (nb >> 3) + (b >> 3) : b;
    const int base = cb * chunk;
    const int end = min(base + chunk, n_px);
    EvalBufs bufs;
    bufs.src = make_rsrc(src0, (unsigned)n_px * 16u);
    bufs.trgP = make_rsrc(lv.trgP, (unsigned)lv.n * 12u);
    bufs.trgD = make_rsrc(lv.trgD, (unsigned)lv.n * 12u);
    bufs.row_bytes = (unsigned)lv.cols * 12u;
    const int i = base + (int)threadIdx.x;
    // ... the first two source records of the span, which depend on nothing, right behind it: in flight during the solve
    SrcCursor cur = {0, 0, 0, 0};
    SrcForm<SRC>::cursor_init(cur, lv, i, kEvalThreads);
    const typename SrcForm<SRC>::T sA = SrcForm<SRC>::load(lv, src0, n_px, base, (unsigned)threadIdx.x << 4, cur);
    const typename SrcForm<SRC>::T sB = SrcForm<SRC>::load(lv, src0, n_px, base + kEvalThreads, (unsigned)threadIdx.x << 4, cur);
    PoseRT T;
    const bool run = fs_solve_and_publish(sh, pend, cfg, level, nb, lv.n, st_out, T);
    if (!run) return;
    const WarpConsts wc = make_warp_consts(T, lv);
    ESTAMP(0);
    PixW wA;
    warp_stage<METHOD, true, SRC>(sA, i < end, T, wc, lv, bufs, wA);
    ESTAMP(1);
    eval_span<METHOD, true, kEvalThreads, SRC>(T, wc, lv, ec, bufs, src0, n_px, base, end, b, nb, wA, sB, cur, partials_out, es, es0);
}

// The tail of a fused-solve schedule: the solve of the last pass enqueued (if one is pending), in place, one block; publishes
// like k_solve.
__global__ __launch_bounds__(kSolveThreads) void k_solve_pending(GNState* st_g, const double* __restrict__ partials, int pend_rows_hint,
                                                                  SolveCfg cfg) {
    __shared__ SolveShared sh;
    constexpr int kStateWords = sizeof(GNState) / 4;
    const int tid = threadIdx.x;
    const int pend = stage_pending(sh, st_g, partials, pend_rows_hint);
    if (pend > 0) {
        SolveCfg c = cfg;
        c.level = sh.sst.level_active;
        c.n_pixels = sh.sst.pend_npix;
        solve_staged(sh, c);
        if (tid == 0) sh.sst.pend_nb = 0;
        __syncthreads();
    }
    if (tid < kStateWords) reinterpret_cast<int*>(st_g)[tid] = reinterpret_cast<const int*>(&sh.sst)[tid];
    if (cfg.host_state) {                                   // uniform
        // (one wave writes the state's ~230 words and runs the one system-scope fence: sixteen waves each asking for a write-back cost
        // the launch 3-4 us)
        if (tid < 64) {
            for (int w = tid; w < kStateWords; w += 64) reinterpret_cast<int*>(cfg.host_state)[w] = reinterpret_cast<const int*>(&sh.sst)[w];
            __threadfence_system();
        }
        __syncthreads();
        if (tid == 0) __hip_atomic_store(cfg.host_tag, cfg.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Standalone GN step for tests: one thread.
struct GnIO {
    float H[36], g[6], pose[16], pose_tmp[16], update[6];
    float lambda;
    int status;
};
__global__ void k_gn_step(GnIO* io) {
    if (threadIdx.x == 0 && blockIdx.x == 0) io->status = gn::step(io->H, io->g, io->lambda, io->pose, io->pose_tmp, io->update);
}

// Warp indices of every source pixel (parity diagnostics).
__global__ void k_warp_indices(LevelDev lv, Pose16 pose, int32_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lv.n) return;
    const PoseRT T = load_pose(pose.v);
    const float4 s = lv.src[i];
    int r = -1, c = -1;
    {       // (no branch on the point's validity around the warp: its predicates are whole-wave lane masks)
        float X, Y, Z, rho2, d2;
        bool vis;
        const WarpConsts wc = {T.tx, T.ty, T.tz, lv.half_nRows, lv.pi_k};
        const unsigned ti = warp_pixel(T, wc, s.x, s.y, s.z, lv, X, Y, Z, rho2, d2, vis);
        vis = vis && s.x != kInvalidPoint;
        if (vis) {
            r = (int)(ti / (unsigned)lv.cols);
            c = (int)ti - r * lv.cols;
        }
    }
    out[2 * i] = r;
    out[2 * i + 1] = c;
}

// Self-test of sqrt_rn / rcp_rn against the compiler's IEEE sqrtf and 1.f/x, and of round_index against
// floor((double)x + 0.5), over a bit-pattern range.
__global__ void k_selftest_math(unsigned first_bits, unsigned count, unsigned long long* __restrict__ mismatches) {
    const unsigned stride = gridDim.x * blockDim.x;
    unsigned bad_s = 0, bad_r = 0, bad_i = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) bad_s += (__builtin_bit_cast(unsigned, sqrt_rn(0.f)) != 0u);      // rho = 0 on the polar axis
    for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += stride) {
        const float x = __builtin_bit_cast(float, first_bits + k);
        const float s0 = sqrtf(x), s1 = sqrt_rn(x);
        const float r0 = 1.f / x, r1 = rcp_rn(x);
        const float q0 = 1.f / -x, q1 = rcp_rn(-x);          // the pinhole warp takes 1 / Z of either sign
        bad_s += (__builtin_bit_cast(unsigned, s0) != __builtin_bit_cast(unsigned, s1));
        bad_r += (__builtin_bit_cast(unsigned, r0) != __builtin_bit_cast(unsigned, r1));
        bad_r += (__builtin_bit_cast(unsigned, q0) != __builtin_bit_cast(unsigned, q1));
        if (fabsf(x) < 1.0e9f) {
            bad_i += (round_index(x) != (int)floor((double)x + 0.5));
            bad_i += (round_index(-x) != (int)floor((double)(-x) + 0.5));
        }
    }
    if (bad_s) atomicAdd(&mismatches[0], (unsigned long long)bad_s);
    if (bad_r) atomicAdd(&mismatches[1], (unsigned long long)bad_r);
    if (bad_i) atomicAdd(&mismatches[2], (unsigned long long)bad_i);
}

// libm_f32.h on the device: asinf_ / atanf_ / roundf_ of the floats first_bits + k, atan2f_ of pairs drawn from k (the host repeats the
// draw: selftest_libm_pair), written out for the host to compare with the C library's own results
__host__ __device__ __forceinline__ void selftest_libm_pair(unsigned k, float& y, float& x) {
    unsigned long long s = 0x9E3779B97F4A7C15ull * (k + 1ull);
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const unsigned long long r = s;
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const unsigned long long q = s;
    unsigned a = (unsigned)r, b = (unsigned)q;
    a = (a & 0x807fffffu) | ((87u + (unsigned)((r >> 32) % 81ull)) << 23);      // binades 2^-40 .. 2^40
    b = (b & 0x807fffffu) | ((87u + (unsigned)((q >> 32) % 81ull)) << 23);
    const unsigned kind = (unsigned)(r >> 40) & 63u;
    if (kind == 0) a &= 0x80000000u;
    if (kind == 1) b &= 0x80000000u;
    if (kind == 2) b = 0x3f800000u;
    if (kind == 3) b = (b & 0x80000000u) | (a & 0x7fffffffu);
    if (kind == 4) b = (b & 0x807fffffu) | (a & 0x7f800000u);
    y = libm32::u2f(a);
    x = libm32::u2f(b);
}
__global__ void k_selftest_libm(unsigned first_bits, unsigned count, float* __restrict__ out /*[4][count]*/) {
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += stride) {
        const float v = libm32::u2f(first_bits + k);
        out[k] = libm32::asinf_(v);
        out[(size_t)count + k] = libm32::atanf_(v);
        out[2 * (size_t)count + k] = libm32::roundf_(v);
        float y, x;
        selftest_libm_pair(first_bits + k, y, x);
        out[3 * (size_t)count + k] = libm32::atan2f_(y, x);
    }
}

// ---------------------------------------------------------------------------------------------------------
// frame preparation
// ---------------------------------------------------------------------------------------------------------
// cv::cvtColor(CV_RGB2GRAY) on 8UC3 (fixed point, shift 14) then convertTo(CV_32FC1, 1./255).
// Four pixels per thread: 12 bytes in as three dwords, 16 bytes out as one float4 (a byte-per-lane version ran at 1.4 TB/s).
// The vector path needs 4-byte aligned rows (base and step); otherwise, and for the last columns, bytes are read one by one.
__device__ __forceinline__ void gray_u8_x4(const uint8_t* __restrict__ rgb, size_t step, int rows, int cols, float* __restrict__ out,
                                           int c4, int r) {
    if (c4 >= cols || r >= rows) return;
    const uint8_t* p = rgb + (size_t)r * step + 3 * (size_t)c4;
    float* o = out + (size_t)r * cols + c4;
    const float k = (float)(1. / 255);
    auto gray = [](unsigned a, unsigned b, unsigned c) { return (int)((4899u * a + 9617u * b + 1868u * c + 8192u) >> 14); };
    if (c4 + 4 <= cols && (((size_t)p | (size_t)o) & 3) == 0 && ((size_t)o & 15) == 0) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
        const uint32_t w0 = q[0], w1 = q[1], w2 = q[2];          // bytes 0..11 = R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
        float4 v;
        v.x = (float)gray(w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u) * k;
        v.y = (float)gray(w0 >> 24, w1 & 255u, (w1 >> 8) & 255u) * k;
        v.z = (float)gray((w1 >> 16) & 255u, w1 >> 24, w2 & 255u) * k;
        v.w = (float)gray((w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24) * k;
        *reinterpret_cast<float4*>(o) = v;
    } else {
        for (int j = 0; j < 4 && c4 + j < cols; ++j) o[j] = (float)gray(p[3 * j], p[3 * j + 1], p[3 * j + 2]) * k;
    }
}

__device__ __forceinline__ void depth_to_f32_x4(const void* __restrict__ depth, size_t step, int depth_type, int rows, int cols,
                                                float* __restrict__ out, int c4, int r) {
    if (c4 >= cols || r >= rows) return;
    const uint8_t* row = (const uint8_t*)depth + (size_t)r * step;
    float* o = out + (size_t)r * cols + c4;
    if (depth_type == 0) {
        const uint16_t* p = (const uint16_t*)row + c4;
        if (c4 + 4 <= cols && ((size_t)p & 7) == 0 && ((size_t)o & 15) == 0) {
            const uint2 w = *reinterpret_cast<const uint2*>(p);
            *reinterpret_cast<float4*>(o) = make_float4((float)(w.x & 0xFFFFu) * 0.001f, (float)(w.x >> 16) * 0.001f,
                                                        (float)(w.y & 0xFFFFu) * 0.001f, (float)(w.y >> 16) * 0.001f);
        } else {
            for (int j = 0; j < 4 && c4 + j < cols; ++j) o[j] = (float)p[j] * 0.001f;
        }
    } else {
        const float* p = (const float*)row + c4;
        for (int j = 0; j < 4 && c4 + j < cols; ++j) o[j] = p[j];
    }
}

// Level 0 of a frame in one launch: blockIdx.z = 0 converts the colour image (cv::cvtColor + convertTo, RPI.h:485-486),
// 1 the depth image (RPI.h:316-319).
__global__ void k_convert_pair(const uint8_t* __restrict__ rgb, size_t rgb_step, const void* __restrict__ depth, size_t depth_step,
                               int depth_type, int rows, int cols, float* __restrict__ gray_out, float* __restrict__ depth_out) {
    const int c4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int r = blockIdx.y;
    if (blockIdx.z == 0) gray_u8_x4(rgb, rgb_step, rows, cols, gray_out, c4, r);
    else depth_to_f32_x4(depth, depth_step, depth_type, rows, cols, depth_out, c4, r);
}

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = (i < 0) ? -i : 2 * n - 2 - i;
    return i;
}

// cv::pyrDown, CV_32FC1, dsize (cols/2, rows/2), BORDER_REFLECT_101; horizontal pass first, then vertical,
// one scale by 1/256 -- same operation order as the oracle.
__device__ __forceinline__ void pyrdown_gray_px(const float* __restrict__ src, int srows, int scols, float* __restrict__ dst,
                                                int dcols, int x, int y) {
    int cc[5], rr[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        cc[k] = reflect101(2 * x - 2 + k, scols);
        rr[k] = reflect101(2 * y - 2 + k, srows);
    }
    float h[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float* row = src + (size_t)rr[k] * scols;
        h[k] = row[cc[2]] * 6 + (row[cc[1]] + row[cc[3]]) * 4 + row[cc[0]] + row[cc[4]];
    }
    const float v = h[2] * 6 + (h[1] + h[3]) * 4 + h[0] + h[4];
    dst[(size_t)y * dcols + x] = v * (1.f / 256.f);
}

// buildPyramidRange level step: mean of the in-range pixels of each 2x2 block, else 0.
__device__ __forceinline__ void pyrdown_depth_px(const float* __restrict__ src, int scols, float* __restrict__ dst, int dcols,
                                                 float min_depth, float max_depth, int x, int y) {
    float av = 0.f;
    unsigned n = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float z = src[(size_t)(2 * y + i) * scols + 2 * x + j];
            if (z > min_depth && z < max_depth) {
                av += z;
                ++n;
            }
        }
    dst[(size_t)y * dcols + x] = n > 0 ? av / n : 0.f;
}

// One pyramid step of BOTH planes of a frame in one launch (blockIdx.z = 0: intensity, 5x5 binomial; 1: depth, valid mean):
// the frame set-up is a chain of small kernels, so its cost is the number of launches, not their bytes.
__global__ void k_pyrdown_pair(const float* __restrict__ gray_src, const float* __restrict__ depth_src, int srows, int scols,
                               float* __restrict__ gray_dst, float* __restrict__ depth_dst, int drows, int dcols, float min_depth,
                               float max_depth) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= dcols || y >= drows) return;
    if (blockIdx.z == 0) pyrdown_gray_px(gray_src, srows, scols, gray_dst, dcols, x, y);
    else pyrdown_depth_px(depth_src, scols, depth_dst, dcols, min_depth, max_depth, x, y);
}

// calcGradientXY on one plane + seam mask; writes the interleaved {value, gradX, gradY} gather record.
__device__ __forceinline__ void gradient_rec_px(const float* __restrict__ src, int rows, int cols, int seam_width,
                                                F3* __restrict__ rec, int r, int c) {
    const float v = src[(size_t)r * cols + c];
    float gx = 0.f, gy = 0.f;
    if (r >= 1 && r < rows - 1 && c >= 1 && c < cols - 1) {
        const float xm = src[(size_t)r * cols + c - 1], xp = src[(size_t)r * cols + c + 1];
        const float ym = src[(size_t)(r - 1) * cols + c], yp = src[(size_t)(r + 1) * cols + c];
        if ((v > xp && v < xm) || (v < xp && v > xm)) gx = 2.f / (1 / (xp - v) + 1 / (v - xm));
        if ((v > yp && v < ym) || (v < yp && v > ym)) gy = 2.f / (1 / (yp - v) + 1 / (v - ym));
    }
    if (seam_width > 1) {   // columns s*w-1 and s*w, s = 1..7: (c + 1) = s*w or s*w + 1
        int s, rem;
        divmod24(c + 1, seam_width, s, rem);
        if (rem <= 1 && s >= 1 && s <= 7) gx = gy = 0.f;
    } else if (seam_width == 1) {
        if (c + 1 >= 1 && c <= 7) gx = gy = 0.f;          // every column 0..7 is s*w-1 or s*w for some s in 1..7
    }
    F3 o;
    o.a = v; o.b = gx; o.c = gy;
    rec[(size_t)r * cols + c] = o;
}

// All gradient records of a target frame (every level, intensity and depth) in ONE launch: a job table in the kernel
// arguments, 256-pixel blocks numbered across the jobs.
constexpr int kMaxJobs = 16;
struct GradJobs {
    const float* src[kMaxJobs];
    F3*          rec[kMaxJobs];
    int rows[kMaxJobs], cols[kMaxJobs], seam[kMaxJobs], first_block[kMaxJobs + 1];
    int n;
};
__global__ void k_gradient_rec_multi(GradJobs jobs) {
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.first_block[j + 1]) ++j;
    const int p = ((int)blockIdx.x - jobs.first_block[j]) * (int)blockDim.x + (int)threadIdx.x;
    const int cols = jobs.cols[j], rows = jobs.rows[j];
    if (p >= rows * cols) return;
    int r, c;
    divmod24(p, cols, r, c);
    gradient_rec_px(jobs.src[j], rows, cols, jobs.seam[j], jobs.rec[j], r, c);
}

// LUT_xyz_sphere + source intensity -> {x,y,z,I}.  sin/cos tables come from the host's libm (same values as
// the CPU path: RPI.h:4559-4571 evaluates them once per column / row).
__device__ __forceinline__ void src_rec_px(const float* __restrict__ depth, const float* __restrict__ gray, int cols,
                                           const float* __restrict__ sin_theta, const float* __restrict__ cos_theta,
                                           const float* __restrict__ sin_phi, const float* __restrict__ cos_phi, float min_depth,
                                           float max_depth, float4* __restrict__ rec, int r, int c) {
    const size_t i = (size_t)r * cols + c;
    const float d = depth[i];
    float4 o;
    o.w = gray[i];
    if (min_depth < d && d < max_depth) {
        o.x = d * sin_phi[r];
        o.y = -d * cos_phi[r] * sin_theta[c];
        o.z = -d * cos_phi[r] * cos_theta[c];
    } else {
        o.x = kInvalidPoint;
        o.y = 0.f;
        o.z = 0.f;
    }
    rec[i] = o;
}
struct SrcJobs {
    const float *depth[8], *gray[8], *sin_theta[8], *cos_theta[8], *sin_phi[8], *cos_phi[8];
    float4* rec[8];
    int rows[8], cols[8], first_block[9];
    int n;
    float min_depth, max_depth;
};
__global__ void k_src_rec_multi(SrcJobs jobs) {
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.first_block[j + 1]) ++j;
    const int p = ((int)blockIdx.x - jobs.first_block[j]) * (int)blockDim.x + (int)threadIdx.x;
    const int cols = jobs.cols[j], rows = jobs.rows[j];
    if (p >= rows * cols) return;
    int r, c;
    divmod24(p, cols, r, c);
    src_rec_px(jobs.depth[j], jobs.gray[j], cols, jobs.sin_theta[j], jobs.cos_theta[j], jobs.sin_phi[j], jobs.cos_phi[j],
               jobs.min_depth, jobs.max_depth, jobs.rec[j], r, c);
}

// ---- lock-step batch forms of the frame set-up kernels (sequence_engine.h): every per-slot plane / record buffer of a level is a
// slice [slot][n] of one allocation; blockIdx.z (or .y) carries the slot; slots outside live_mask have no frame this round ----
constexpr int kMaxSlots = 32;
struct FramePtrs {
    const uint8_t* rgb[kMaxSlots];
    const void* depth[kMaxSlots];
};
__global__ void k_convert_pair_b(FramePtrs fp, size_t rgb_step, size_t depth_step, int depth_type, int rows, int cols,
                                 float* __restrict__ gray_out, float* __restrict__ depth_out, unsigned long long live_mask) {
    const int slot = blockIdx.z >> 1;
    if (!((live_mask >> slot) & 1ull)) return;
    const int c4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int r = blockIdx.y;
    const size_t off = (size_t)slot * (size_t)rows * (size_t)cols;
    if ((blockIdx.z & 1) == 0) gray_u8_x4(fp.rgb[slot], rgb_step, rows, cols, gray_out + off, c4, r);
    else depth_to_f32_x4(fp.depth[slot], depth_step, depth_type, rows, cols, depth_out + off, c4, r);
}
__global__ void k_pyrdown_pair_b(const float* __restrict__ gray_src, const float* __restrict__ depth_src, int srows, int scols,
                                 float* __restrict__ gray_dst, float* __restrict__ depth_dst, int drows, int dcols, float min_depth,
                                 float max_depth, unsigned long long live_mask) {
    const int slot = blockIdx.z >> 1;
    if (!((live_mask >> slot) & 1ull)) return;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= dcols || y >= drows) return;
    const size_t so = (size_t)slot * (size_t)srows * (size_t)scols, dof = (size_t)slot * (size_t)drows * (size_t)dcols;
    if ((blockIdx.z & 1) == 0) pyrdown_gray_px(gray_src + so, srows, scols, gray_dst + dof, dcols, x, y);
    else pyrdown_depth_px(depth_src + so, scols, depth_dst + dof, dcols, min_depth, max_depth, x, y);
}
__global__ void k_gradient_rec_multi_b(GradJobs jobs, unsigned long long live_mask) {
    const int slot = blockIdx.y;
    if (!((live_mask >> slot) & 1ull)) return;
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.first_block[j + 1]) ++j;
    const int p = ((int)blockIdx.x - jobs.first_block[j]) * (int)blockDim.x + (int)threadIdx.x;
    const int cols = jobs.cols[j], rows = jobs.rows[j];
    if (p >= rows * cols) return;
    int r, c;
    divmod24(p, cols, r, c);
    const size_t off = (size_t)slot * (size_t)rows * (size_t)cols;
    gradient_rec_px(jobs.src[j] + off, rows, cols, jobs.seam[j], jobs.rec[j] + off, r, c);
}
__global__ void k_src_rec_multi_b(SrcJobs jobs, unsigned long long live_mask) {
    const int slot = blockIdx.y;
    if (!((live_mask >> slot) & 1ull)) return;
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.first_block[j + 1]) ++j;
    const int p = ((int)blockIdx.x - jobs.first_block[j]) * (int)blockDim.x + (int)threadIdx.x;
    const int cols = jobs.cols[j], rows = jobs.rows[j];
    if (p >= rows * cols) return;
    int r, c;
    divmod24(p, cols, r, c);
    const size_t off = (size_t)slot * (size_t)rows * (size_t)cols;
    src_rec_px(jobs.depth[j] + off, jobs.gray[j] + off, cols, jobs.sin_theta[j], jobs.cos_theta[j], jobs.sin_phi[j], jobs.cos_phi[j],
               jobs.min_depth, jobs.max_depth, jobs.rec[j] + off, r, c);
}

// ---------------------------------------------------------------------------------------------------------
// Fused frame set-up of the sequence engine: ONE launch per pyramid level takes the level's input (level 0: the raw colour /
// depth images; level >= 1: the float planes the previous launch left) through an LDS tile and writes everything the alignment
// reads from this level -- the source records {x,y,z,I} (RPI.h:4554-4587), the target records {v,gx,gy} of intensity and depth
// (calcGradientXY + seam mask, RPI.h:365-398, 4538-4549) -- plus the next level's two planes (cv::pyrDown / buildPyramidRange,
// RPI.h:292-354).  The level-0 float planes are never stored: 5 B/px in, 40 B/px of records out, against 8 B/px written and
// 24 B/px re-read by the four separate kernels.  Every value is produced by the same float operations in the same order as
// gray_u8_x4 / depth_to_f32_x4 / pyrdown_gray_px / pyrdown_depth_px / gradient_rec_px / src_rec_px: records are bit-identical.
// Tile = 64 x 16 pixels + a 2-pixel ring (the 5x5 binomial of the next level's pixels reaches 2 pixels out; BORDER_REFLECT_101
// is applied when the ring is loaded, so the window code has no border case).  blockIdx.z = slot.
// ---------------------------------------------------------------------------------------------------------
constexpr int kFsTW = 64, kFsTH = 16, kFsRing = 2;
constexpr int kFsLW = kFsTW + 2 * kFsRing, kFsLH = kFsTH + 2 * kFsRing;      // 68 x 20
constexpr int kFsRawDw = (kFsLW * 3 + 3 + 3) / 4 + 1;                        // dwords that cover 204 bytes at any byte alignment
struct FrameLevelArgs {
    int rows, cols;                 // this level
    int drows, dcols;               // next level (0: none)
    int seam;                       // seam-mask width (cols / 8) or 0
    int depth_type;                 // level 0: 0 = u16 mm, 1 = f32 m
    size_t rgb_step, depth_step;    // level 0: bytes per row of the raw images
    const float *gray_in, *depth_in;      // level >= 1: [slot][n] planes
    float *gray_next, *depth_next;        // [slot][drows * dcols]
    float4* src_rec;                // [slot][n]
    F3 *trg_p, *trg_d;              // [slot][n]
    const float *sin_theta, *cos_theta, *sin_phi, *cos_phi;
    float min_depth, max_depth;
    unsigned long long live_mask;   // slots with a frame in this launch
    unsigned long long src_mask;    // ... whose source records are wanted
    unsigned long long trg_mask;    // ... whose target records are wanted
    int compact_src;                // spherical source records as {depth, Isrc} (8 B: the pass re-forms the point, SrcForm<2>) instead of {x, y, z, Isrc}
    int pinhole;                    // source records of a pinhole sensor (RPI.h:4277-4300) instead of the spherical LUT
    float pin_ox, pin_oy, pin_inv_fx, pin_inv_fy;      // intrinsics of THIS level
};

// 1.f / x, bit for bit: inside [2^-60, 2^60] the 3-instruction rcp_rn (proven equal to the IEEE quotient over that whole range by
// rgbd360_selftest_math), the compiler's division sequence (~12 instructions) elsewhere -- a branch no real image takes.
__device__ __forceinline__ float rcp_ieee(float x) {
    const float ax = fabsf(x);
    if (ax >= 0x1p-60f && ax <= 0x1p60f) return rcp_rn(x);
    return 1.f / x;
}
// calcGradientXY's harmonic mean of the two one-sided differences, 2 / (1 / a + 1 / b) (RPI.h:381-390): a and b have the same
// sign here, 2 * RN(1 / s) == RN(2 / s) (scaling by 2 is exact), so this equals gradient_rec_px's three IEEE divisions bit for bit.
__device__ __forceinline__ float harmonic2(float a, float b) { return 2.f * rcp_ieee(rcp_ieee(a) + rcp_ieee(b)); }

template <bool RAW>
__global__ __launch_bounds__(256) void k_frame_level_b(FrameLevelArgs A, FramePtrs fp) {
    const int slot = blockIdx.z;
    if (!((A.live_mask >> slot) & 1ull)) return;
    __shared__ float sg[kFsLH][kFsLW], sd[kFsLH][kFsLW];
    __shared__ uint32_t raw[RAW ? kFsLH : 1][RAW ? kFsRawDw : 1];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * kFsTW, r0 = blockIdx.y * kFsTH;
    const int rows = A.rows, cols = A.cols;
    const size_t n = (size_t)rows * (size_t)cols;
    constexpr int kPerThread = (kFsLH * kFsLW + 255) / 256;      // 6 tile pixels per thread in the load phase
    if (RAW) {
        const uint8_t* rgb = fp.rgb[slot];
        const uint8_t* dep = (const uint8_t*)fp.depth[slot];
        const float k255 = (float)(1. / 255);
        auto gray = [](unsigned a, unsigned b, unsigned c) { return (int)((4899u * a + 9617u * b + 1868u * c + 8192u) >> 14); };
        // Colour rows travel as raw dwords: a 3-byte pixel stream has no natural alignment, so the dwords covering the tile's
        // 68-pixel column span [cs, cs + 68) are loaded and the bytes are picked out of LDS.  The span is clamped into the image
        // (at the left / right border the reflected ring columns lie inside it), ring rows are loaded from their reflected row.
        // Not for tiles that reach the last image row (a covering dword may extend 3 bytes past a row's pixels) or images
        // narrower than the span: those read byte by byte.
        const bool fast = cols >= kFsLW && r0 + kFsTH + kFsRing < rows && (r0 > 0 || ((size_t)rgb & 3) == 0);
        const int cs = min(max(c0 - kFsRing, 0), cols - kFsLW);
        float dv[kPerThread];
        if (fast) {
            uint32_t rw[(kFsLH * kFsRawDw + 255) / 256];
#pragma unroll
            for (int q = 0; q < (kFsLH * kFsRawDw + 255) / 256; ++q) {
                const int i = tid + q * 256;
                const int ly = min(i / kFsRawDw, kFsLH - 1), k = i - (i / kFsRawDw) * kFsRawDw;
                const int r = reflect101(r0 - kFsRing + ly, rows);
                const size_t addr = (size_t)(rgb + (size_t)r * A.rgb_step + 3 * (size_t)cs);
                rw[q] = *reinterpret_cast<const uint32_t*>((addr & ~(size_t)3) + 4 * (size_t)k);
            }
#pragma unroll
            for (int q = 0; q < kPerThread; ++q) {
                const int i = min(tid + q * 256, kFsLH * kFsLW - 1);
                const int ly = i / kFsLW, lx = i - ly * kFsLW;
                const int r = reflect101(r0 - kFsRing + ly, rows), c = reflect101(c0 - kFsRing + lx, cols);
                const uint8_t* drow = dep + (size_t)r * A.depth_step;
                dv[q] = A.depth_type == 0 ? (float)((const uint16_t*)drow)[c] * 0.001f : ((const float*)drow)[c];
            }
#pragma unroll
            for (int q = 0; q < (kFsLH * kFsRawDw + 255) / 256; ++q) {
                const int i = tid + q * 256;
                if (i < kFsLH * kFsRawDw) raw[i / kFsRawDw][i - (i / kFsRawDw) * kFsRawDw] = rw[q];
            }
#pragma unroll
            for (int q = 0; q < kPerThread; ++q) {
                const int i = tid + q * 256;
                if (i < kFsLH * kFsLW) sd[i / kFsLW][i - (i / kFsLW) * kFsLW] = dv[q];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < kPerThread; ++q) {
                const int i = tid + q * 256;
                if (i < kFsLH * kFsLW) {
                    const int ly = i / kFsLW, lx = i - ly * kFsLW;
                    const int r = reflect101(r0 - kFsRing + ly, rows), c = reflect101(c0 - kFsRing + lx, cols);
                    const size_t addr = (size_t)(rgb + (size_t)r * A.rgb_step + 3 * (size_t)cs);
                    const uint8_t* px = reinterpret_cast<const uint8_t*>(&raw[ly][0]) + (addr & 3) + 3 * (c - cs);
                    sg[ly][lx] = (float)gray(px[0], px[1], px[2]) * k255;
                }
            }
        } else {
            for (int i = tid; i < kFsLH * kFsLW; i += 256) {
                const int ly = i / kFsLW, lx = i - ly * kFsLW;
                const int r = reflect101(r0 - kFsRing + ly, rows), c = reflect101(c0 - kFsRing + lx, cols);
                const uint8_t* px = rgb + (size_t)r * A.rgb_step + 3 * (size_t)c;
                sg[ly][lx] = (float)gray(px[0], px[1], px[2]) * k255;
                const uint8_t* drow = dep + (size_t)r * A.depth_step;
                sd[ly][lx] = A.depth_type == 0 ? (float)((const uint16_t*)drow)[c] * 0.001f : ((const float*)drow)[c];
            }
        }
    } else {
        const float* gin = A.gray_in + (size_t)slot * n;
        const float* din = A.depth_in + (size_t)slot * n;
        float gv[kPerThread], dv[kPerThread];
#pragma unroll
        for (int q = 0; q < kPerThread; ++q) {      // all loads of the tile first, then the LDS writes: one memory round trip
            const int i = min(tid + q * 256, kFsLH * kFsLW - 1);
            const int ly = i / kFsLW, lx = i - ly * kFsLW;
            const int r = reflect101(r0 - kFsRing + ly, rows), c = reflect101(c0 - kFsRing + lx, cols);
            gv[q] = gin[(size_t)r * cols + c];
            dv[q] = din[(size_t)r * cols + c];
        }
#pragma unroll
        for (int q = 0; q < kPerThread; ++q) {
            const int i = tid + q * 256;
            if (i < kFsLH * kFsLW) {
                sg[i / kFsLW][i - (i / kFsLW) * kFsLW] = gv[q];
                sd[i / kFsLW][i - (i / kFsLW) * kFsLW] = dv[q];
            }
        }
    }
    __syncthreads();

    // ---- records of the tile.  A thread owns the pixels j, j + 16, j + 32, j + 48 of one tile row (16 threads per row): every
    //      store instruction then writes, per row, 16 consecutive records -- 256 B of source records, 192 B of target records, whole
    //      64-byte segments -- instead of 64 lanes each dropping 16 bytes into a different segment (which ran at 1.8 TB/s) ----
    {
        const int ty = tid >> 4, tj = tid & 15;
        const int r = r0 + ty;
        if (r < rows) {
            const bool want_src = (A.src_mask >> slot) & 1ull, want_trg = (A.trg_mask >> slot) & 1ull;
            const int ly = ty + kFsRing;
            if (want_src && A.compact_src) {       // {depth, Isrc}: validity and direction are the pass's business (src_point)
                float2* out = reinterpret_cast<float2*>(A.src_rec) + (size_t)slot * n + (size_t)r * cols;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = c0 + tj + 16 * k;
                    if (c < cols) out[c] = make_float2(sd[ly][tj + 16 * k + kFsRing], sg[ly][tj + 16 * k + kFsRing]);
                }
            } else if (want_src) {
                const float sp = A.pinhole ? 0.f : A.sin_phi[r], cp = A.pinhole ? 0.f : A.cos_phi[r];
                float4* out = A.src_rec + (size_t)slot * n + (size_t)r * cols;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = c0 + tj + 16 * k;
                    if (c < cols) {
                        const int lx = tj + 16 * k + kFsRing;
                        const float d = sd[ly][lx];
                        float4 o;
                        o.w = sg[ly][lx];
                        if (A.pinhole) {                   // k_src_rec_pinhole's arithmetic: z is stored for every pixel
                            o.z = d;
                            if (A.min_depth < d && d < A.max_depth) {
                                o.x = ((float)c - A.pin_ox) * d * A.pin_inv_fx;
                                o.y = ((float)r - A.pin_oy) * d * A.pin_inv_fy;
                            } else {
                                o.x = kInvalidPoint;
                                o.y = 0.f;
                            }
                        } else if (A.min_depth < d && d < A.max_depth) {
                            o.x = d * sp;
                            o.y = -d * cp * A.sin_theta[c];
                            o.z = -d * cp * A.cos_theta[c];
                        } else {
                            o.x = kInvalidPoint;
                            o.y = 0.f;
                            o.z = 0.f;
                        }
                        out[c] = o;
                    }
                }
            }
            if (want_trg) {
                F3* recP = A.trg_p + (size_t)slot * n + (size_t)r * cols;
                F3* recD = A.trg_d + (size_t)slot * n + (size_t)r * cols;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = c0 + tj + 16 * k;
                    if (c < cols) {
                        const int lx = tj + 16 * k + kFsRing;
                        bool seam = false;      // seam-mask columns (RPI.h:4538-4549), shared by both planes
                        if (A.seam > 1) {
                            int sidx, rem;
                            divmod24(c + 1, A.seam, sidx, rem);
                            seam = rem <= 1 && sidx >= 1 && sidx <= 7;
                        } else if (A.seam == 1) {
                            seam = c + 1 >= 1 && c <= 7;
                        }
                        const bool inner = r >= 1 && r < rows - 1 && c >= 1 && c < cols - 1;
#pragma unroll
                        for (int plane = 0; plane < 2; ++plane) {
                            const float(*S)[kFsLW] = plane == 0 ? sg : sd;
                            const float v = S[ly][lx];
                            float gx = 0.f, gy = 0.f;
                            if (inner) {
                                const float xm = S[ly][lx - 1], xp = S[ly][lx + 1];
                                const float ym = S[ly - 1][lx], yp = S[ly + 1][lx];
                                if ((v > xp && v < xm) || (v < xp && v > xm)) gx = harmonic2(xp - v, v - xm);
                                if ((v > yp && v < ym) || (v < yp && v > ym)) gy = harmonic2(yp - v, v - ym);
                            }
                            if (seam) gx = gy = 0.f;
                            F3 o;
                            o.a = v; o.b = gx; o.c = gy;
                            (plane == 0 ? recP : recD)[c] = o;
                        }
                    }
                }
            }
        }
    }
    // ---- the next level's planes: one output pixel per thread (32 x 8 per tile) ----
    if (A.drows > 0) {
        const int oy = tid >> 5, ox = tid & 31;
        const int x = (c0 >> 1) + ox, y = (r0 >> 1) + oy;
        if (x < A.dcols && y < A.drows) {
            const size_t dn = (size_t)A.drows * (size_t)A.dcols;
            const int ly = 2 * oy, lx = 2 * ox;      // window origin (2y - 2, 2x - 2) in tile coordinates
            float h[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const float* row = &sg[ly + k][lx];
                h[k] = row[2] * 6 + (row[1] + row[3]) * 4 + row[0] + row[4];
            }
            const float v = h[2] * 6 + (h[1] + h[3]) * 4 + h[0] + h[4];
            A.gray_next[(size_t)slot * dn + (size_t)y * A.dcols + x] = v * (1.f / 256.f);
            float av = 0.f;
            unsigned cnt = 0;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float z = sd[ly + kFsRing + i][lx + kFsRing + j];
                    if (z > A.min_depth && z < A.max_depth) {
                        av += z;
                        ++cnt;
                    }
                }
            A.depth_next[(size_t)slot * dn + (size_t)y * A.dcols + x] = cnt > 0 ? av / cnt : 0.f;
        }
    }
}




}  // namespace r360
