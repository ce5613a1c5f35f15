// Pinhole single-sensor alignment: RegisterPhotoICP::alignFrames RPI.h:4254-4512 with
// errorPhotoICP RPI.h:560-748 and calcHessGrad RPI.h:754-1104 (occlusion 0; bUseSalientPixels through sal_thr), and the occlusion-aware
// variants errorPhotoICP_Occ1/2, calcHessGrad_Occ1/2 RPI.h:1107-2030 (second half of this file).
//
// The sensor images are small (320x240 per Asus sensor of the rig), so these passes are launch- and latency-bound, not
// bandwidth-bound; the kernels are the plain form of k_eval (no software pipelining), one fused pass producing
//   * the error sums of errorPhotoICP (every visible pixel, NO saliency test; both averages / nValidDepthPts), and
//   * the normal equations of calcHessGrad (saliency-gated rows; a flat depth gradient drops the photometric row too)
// at one pose, in k_eval's partial-row layout so that k_solve's reduction mode is shared.  The Levenberg-Marquardt
// driver runs on the host (one state read-back per evaluation).
#pragma once
#include "photo_icp_kernels.h"

namespace r360 {

struct PinK {
    float fx, fy, ox, oy;     // intrinsics of the pyramid level (RPI.h:571-575)
};

// RPI.h:4277-4300: source record {x, y, z, Isrc}; x = -10000 marks a depth outside (min_depth, max_depth)
__global__ void k_src_rec_pinhole(const float* __restrict__ depth, const float* __restrict__ gray, int rows, int cols, PinK K,
                                  float inv_fx, float inv_fy, float min_depth, float max_depth, float4* __restrict__ rec) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (c >= cols || r >= rows) return;
    const size_t i = (size_t)r * cols + c;
    const float z = depth[i];
    float4 o;
    o.z = z;
    o.w = gray[i];
    if (min_depth < z && z < max_depth) {
        o.x = ((float)c - K.ox) * z * inv_fx;
        o.y = ((float)r - K.oy) * z * inv_fy;
    } else {
        o.x = kInvalidPoint;
        o.y = 0.f;
    }
    rec[i] = o;
}

// Device arithmetic definition of the pinhole warp (the oracle's math_mode 1 repeats it): fma rotation, correctly rounded
// 1/Z, column = round(fma(X fx, 1/Z, ox)), row = round(fma(Y fy, 1/Z, oy)), round = floor(x + 0.5).
// libm != 0 (rgbd360_set_index_arithmetic(ctx, 1)): the REFERENCE's arithmetic instead, RPI.h:701-708 as compiled -- Eigen's product order
// without fused multiply-adds, inv_z = 1.0 / Z in double stored to float, (X fx) inv_z + ox, roundf, the x86 float -> int conversion; the
// oracle's math_mode 0.
__device__ __forceinline__ unsigned warp_pinhole(const PoseRT& T, float px, float py, float pz, const PinK& K, int rows, int cols,
                                                 float& X, float& Y, float& Z, float& inv_z, bool& vis, int libm = 0) {
    if (libm) {                                       // uniform (a kernel argument)
        X = ((T.r00 * px + T.r01 * py) + T.r02 * pz) + T.tx;
        Y = ((T.r10 * px + T.r11 * py) + T.r12 * pz) + T.ty;
        Z = ((T.r20 * px + T.r21 * py) + T.r22 * pz) + T.tz;
        inv_z = (float)(1.0 / (double)Z);
        const float tc = (X * K.fx) * inv_z + K.ox;
        const float tr = (Y * K.fy) * inv_z + K.oy;
        const bool fin = isfinite(tr) && isfinite(tc);
        const float fr = libm32::roundf_(fin ? tr : -1.f), fc = libm32::roundf_(fin ? tc : -1.f);
        const int ri = (fr >= -2147483648.f && fr < 2147483648.f) ? (int)fr : (int)0x80000000;
        const int ci = (fc >= -2147483648.f && fc < 2147483648.f) ? (int)fc : (int)0x80000000;
        vis = fin && ((unsigned)ri < (unsigned)rows) && ((unsigned)ci < (unsigned)cols);
        return (unsigned)(ri * cols + ci);
    }
    X = fmaf(T.r02, pz, fmaf(T.r01, py, fmaf(T.r00, px, T.tx)));
    Y = fmaf(T.r12, pz, fmaf(T.r11, py, fmaf(T.r10, px, T.ty)));
    Z = fmaf(T.r22, pz, fmaf(T.r21, py, fmaf(T.r20, px, T.tz)));
    inv_z = rcp_rn(Z);
    const float tc = fmaf(X * K.fx, inv_z, K.ox);
    const float tr = fmaf(Y * K.fy, inv_z, K.oy);
    const bool sane = (fabsf(tr) < 1e9f) && (fabsf(tc) < 1e9f);
    const int ri = round_index(sane ? tr : -1.f), ci = round_index(sane ? tc : -1.f);
    vis = sane && ((unsigned)ri < (unsigned)rows) && ((unsigned)ci < (unsigned)cols);
    return (unsigned)(ri * cols + ci);
}

template <int METHOD>
// The pose of the pass arrives as a kernel argument: the Levenberg-Marquardt loop lives on the host and evaluates one pose per
// round trip, so there is no device state to gate on and no initialisation launch in front of the pass.
// sal_thr >= 0: useSaliency(true) (RPI.h:266, 590-690) -- the ERROR sums run over vSalientPixels only, the interior pixels whose
// TARGET gray gradient exceeds thresSaliency in x or y (RPI.h:420-424), used as SOURCE pixel indices (RPI.h:613-634: as written);
// the normal equations keep every pixel (calcHessGrad's salient branch is commented out, RPI.h:813-870).  Border gradients are
// zero (RPI.h:367-372), so the list's "interior" condition is the threshold test itself.
__global__ __launch_bounds__(kEvalThreads) void k_eval_pinhole(LevelDev lv, PinK K, EvalConsts ec, Pose16 pose,
                                                                double* __restrict__ partials, int chunk, int level, float sal_thr) {
    const int b = blockIdx.x;
    const int base = b * chunk;
    const int end = min(base + chunk, lv.n);
    const PoseRT T = load_pose(pose.v);

    EvalAcc A;
#pragma unroll
    for (int k = 0; k < 27; ++k) A.acc[k] = 0.f;
    A.e2p = A.e2d = 0.f;
    A.nP = A.nD = A.nVis = 0;

    const int n_steps = (end - base + kEvalThreads - 1) / kEvalThreads;      // wave-uniform: the ballots count whole waves
    for (int k = 0; k < n_steps; ++k) {
        const int i = base + k * kEvalThreads + (int)threadIdx.x;
        const bool in_range = i < end;
        const float4 s = lv.src[in_range ? i : lv.n - 1];
        float X, Y, Z, iz;
        bool vis;
        unsigned ti = warp_pinhole(T, s.x, s.y, s.z, K, lv.rows, lv.cols, X, Y, Z, iz, vis, lv.libm);
        vis = vis && in_range && (s.x != kInvalidPoint);
        ti = vis ? ti : 0u;
        bool evis = vis;                  // the pixel takes part in the error sums
        if (sal_thr >= 0.f) {             // uniform
            const F3 ts = lv.trgP[in_range ? i : 0];
            evis = vis && (fabsf(ts.b) > sal_thr || fabsf(ts.c) > sal_thr);
        }
        F3 tp = {0.f, 0.f, 0.f}, td = {0.f, 0.f, 0.f};
        if (METHOD != 1) tp = lv.trgP[ti];
        if (METHOD != 0) td = lv.trgD[ti];
        const float depth2 = td.a;
        const bool sal_p = !(fabsf(tp.b) < ec.thr_photo && fabsf(tp.c) < ec.thr_photo);
        const bool sal_d = !(fabsf(td.b) < ec.thr_depth && fabsf(td.c) < ec.thr_depth);
        const float iz2 = iz * iz;

        if (METHOD != 1) {
            A.nP += ballot_count(evis);                                           // errorPhotoICP: no saliency test (RPI.h:712-721)
            const bool row_on = vis && sal_p && (METHOD == 0 || sal_d);           // calcHessGrad: RPI.h:906-907, 929-930
            A.nVis += ballot_count(row_on);
            if (vis) {
#pragma clang fp contract(fast)
                const float photoDiff = tp.a - s.w;
                const float wpf = weight_huber_fast(photoDiff, ec.sigma_photo) * ec.sigma_photo_inv_f;
                const float res = wpf * photoDiff;
                A.e2p += evis ? res * res : 0.f;
                if (row_on) {
                    const float wgx = wpf * tp.b * K.fx, wgy = wpf * tp.c * K.fy;
                    accumulate_row(A, wgx * iz, wgy * iz, -(wgx * X + wgy * Y) * iz2, X, Y, Z, res);
                }
            }
        }
        if (METHOD != 0) {
            const bool err_on = vis && isfinite(depth2);                          // RPI.h:722-735
            A.nD += ballot_count(err_on && evis);
            const bool row_on = err_on && sal_d && (METHOD == 1 || sal_p);
            A.nVis += ballot_count(row_on);
            if (err_on) {
#pragma clang fp contract(fast)
                const float depthDiff = depth2 - Z;
                const float sd = ec.sigma_depth * Z;
                const float wd = weight_huber_fast(depthDiff, sd) * fast_rcp(sd);
                const float res = wd * depthDiff;
                A.e2d += evis ? res * res : 0.f;
                if (row_on) {
                    const float gx = td.b * K.fx, gy = td.c * K.fy;
                    accumulate_row(A, wd * (gx * iz), wd * (gy * iz), wd * (-(gx * X + gy * Y) * iz2 - 1.f), X, Y, Z, res);
                }
            }
        }
    }

    __shared__ double red[kEvalThreads / 64][kNumPartials];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        float v[32], out[2];
#pragma unroll
        for (int k = 0; k < 27; ++k) v[k] = A.acc[k];
        v[P_E2P] = A.e2p;
        v[P_E2D] = A.e2d;
        v[P_NP] = v[P_ND] = v[P_NVIS] = 0.f;
        wave_reduce32(v, out);
        if ((lane & 3) == 0) {
            const int row = lane >> 4, quad = (lane >> 2) & 3;
            const int idx = 2 * (quad & 1) + 4 * (quad >> 1) + 8 * (row & 1) + 16 * (row >> 1);
            if (idx + 0 < P_NP) red[wave][idx + 0] = (double)out[0];
            if (idx + 1 < P_NP) red[wave][idx + 1] = (double)out[1];
        }
        if (lane == 63) {
            red[wave][P_NP] = (double)A.nP;
            red[wave][P_ND] = (double)A.nD;
            red[wave][P_NVIS] = (double)A.nVis;
        }
    }
    __syncthreads();
    if (threadIdx.x < kNumPartials) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < kEvalThreads / 64; ++w) v += red[w][threadIdx.x];
        partials[(size_t)b * kNumPartials + threadIdx.x] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Occlusion-aware pinhole passes: errorPhotoICP_Occ1 / calcHessGrad_Occ1 (RPI.h:1107-1544), errorPhotoICP_Occ2 /
// calcHessGrad_Occ2 (RPI.h:1547-2030).  Sequential semantics of the source (its OpenMP loops race on the z-buffer): per TARGET
// pixel, the source pixels that land on it are visited in index order against a z-buffer entry
//     error pass   skip if (buf > 0 && 1/Z < buf), else buf = 1/Z; the residual slot of the TARGET pixel is overwritten by every
//                  accepted pixel that passes the saliency tests, the counters count every such write
//     H, g pass    first arrival counts twice in numVisiblePixels (as written), later ones are skipped if 1/Z < buf; rows are kept
//                  per SOURCE pixel (no retraction) and summed where the PHOTOMETRIC residual is non-zero -- the depth rows too
//                  (as written: DEPTH_CONSISTENCY alone gives H = 0)
//     Occ2         an outlier gate in front: |Dtrg - 1/Z| > 1 m in the error pass (sic), |Dtrg - Z| > 1 m in the H, g pass
// That is a state machine along each target pixel's list of source pixels IN INDEX ORDER.  No sort: k_pin_occ_keys counts the arrivals
// per target pixel (integer atomics: order-independent) and keeps the first kPinShort arrivals of a target in its slot row (in arrival
// order, i.e. unordered); an arrival beyond those also widens the target's BOX -- lowest / highest source index and lowest / highest
// source column of the late arrivals (four more atomics, only on the targets that need them).  k_pin_occ_walk gives every TARGET pixel
// to one thread, which visits its list in index order -- a short list by repeated selection of the next larger index among its
// <= kPinShort slots, a long one (zoom-outs and collapses pile tens to tens of thousands of source pixels on one target) by scanning the
// key array over the box of ALL its arrivals (the late ones' box widened by the slots), row by row: the source pixels of one target form
// a compact patch, so the scan reads about as many keys as the list is long (scanning every key between the lowest and the highest
// index read whole image rows for a 5 x 5 patch: 0.85 ms instead of 0.08 per 640 x 480 evaluation at a 3 m zoom-out).  Exact sequential
// semantics whatever the list lengths, no residency or ordering assumption, no library (rounds 3-4 sorted the (target, source) pairs
// with a library radix sort: 8-10 launches, 44 us per evaluation).  A collapse of the whole image onto a few pixels stays what its
// semantics make it: one thread per target walking tens of thousands of arrivals (tens of milliseconds; tools/pinhole_occ_longlist_perf.py).
// One fused walk yields the error sums and the normal equations at the pose, in k_eval's partial-row layout; it also re-arms the
// per-target words.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr float kPinThresDepthOutliers = 1.f;      // thresDepthOutliers = maxDepthOutliers (RPI.h:215, 4256-4260)
constexpr unsigned kPinOccErr = 1u << 30, kPinOccHess = 1u << 31, kPinOccIndex = 0xFFFFFFu;

constexpr int kPinShort = 8;                       // arrivals per target pixel kept in its slot row
struct PinOccLists {                               // per target pixel of the largest level (armed once at allocation, re-armed by the walk)
    int* cnt;                                      // arrivals (0)
    int4* box;                                     // of the arrivals beyond the slot row: {lowest index, highest index, lowest column, highest column}
                                                   // (armed {INT_MAX, -1, INT_MAX, -1})
    unsigned* slots;                               // [n][kPinShort] source indices, arrival order
};
constexpr int4 kPinBoxArmed = {0x7fffffff, -1, 0x7fffffff, -1};
__global__ void k_pin_occ_arm(int* __restrict__ cnt, int4* __restrict__ box, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        cnt[i] = 0;
        box[i] = kPinBoxArmed;
    }
}
template <int OCC>
__global__ __launch_bounds__(256) void k_pin_occ_keys(LevelDev lv, PinK K, Pose16 pose, unsigned* __restrict__ keys, unsigned* __restrict__ vals, PinOccLists Ls) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lv.n) return;
    const PoseRT T = load_pose(pose.v);
    const float4 s = lv.src[i];
    float X, Y, Z, iz;
    bool vis;
    const unsigned ti = warp_pinhole(T, s.x, s.y, s.z, K, lv.rows, lv.cols, X, Y, Z, iz, vis, lv.libm);
    const bool cand = vis && (s.x != kInvalidPoint);
    bool cand_e = cand, cand_h = cand;
    if (OCC == 2 && cand) {
        const float depth2 = lv.trgD[ti].a;
        cand_e = !(fabsf(depth2 - iz) > kPinThresDepthOutliers);       // RPI.h:1687-1690 (depth against inverse depth: as written)
        cand_h = !(fabsf(depth2 - Z) > kPinThresDepthOutliers);        // RPI.h:1857-1862
    }
    const bool in_list = cand_e || cand_h;
    keys[i] = in_list ? ti : (unsigned)lv.n;                            // (n: on no list)
    vals[i] = (unsigned)i | (cand_e ? kPinOccErr : 0u) | (cand_h ? kPinOccHess : 0u);
    if (in_list) {
        const int slot = atomicAdd(&Ls.cnt[ti], 1);
        if (slot < kPinShort) Ls.slots[(size_t)ti * kPinShort + slot] = (unsigned)i;
        else {                                                          // a long list: the walk scans the box of its arrivals
            int* b = reinterpret_cast<int*>(&Ls.box[ti]);
            const int c = i % lv.cols;
            atomicMin(&b[0], i);
            atomicMax(&b[1], i);
            atomicMin(&b[2], c);
            atomicMax(&b[3], c);
        }
    }
}

constexpr int kPinWalkThreads = 256;
template <int METHOD>
__global__ __launch_bounds__(kPinWalkThreads) void k_pin_occ_walk(LevelDev lv, PinK K, EvalConsts ec, Pose16 pose, const unsigned* __restrict__ keys,
                                                                   const unsigned* __restrict__ vals, PinOccLists Ls, double* __restrict__ partials) {
    const int p = blockIdx.x * kPinWalkThreads + (int)threadIdx.x;      // the target pixel of this thread
    const PoseRT T = load_pose(pose.v);
    EvalAcc A;
#pragma unroll
    for (int k = 0; k < 27; ++k) A.acc[k] = 0.f;
    A.e2p = A.e2d = 0.f;
    int nP = 0, nD = 0, nVis = 0;
    int m = 0;
    int4 box = kPinBoxArmed;
    if (p < lv.n) {
        m = Ls.cnt[p];
        if (m) Ls.cnt[p] = 0;                                          // re-armed for the next evaluation
        if (m > kPinShort) {
            box = Ls.box[p];
            Ls.box[p] = kPinBoxArmed;
        }
    }
    if (m) {
        const unsigned ti = (unsigned)p;
        F3 tp = {0.f, 0.f, 0.f};
        if (METHOD != 1) tp = lv.trgP[ti];
        const F3 td = lv.trgD[ti];
        const float depth2 = td.a;
        const bool sal_p = !(fabsf(tp.b) < ec.thr_photo && fabsf(tp.c) < ec.thr_photo);
        const bool sal_d = !(fabsf(td.b) < ec.thr_depth && fabsf(td.c) < ec.thr_depth);
        const bool fin_d = isfinite(depth2);
        float buf_e = 0.f, buf_h = 0.f;              // invDepthBuffer(ii) of the two passes
        float res_p = 0.f, res_d = 0.f;              // residualsPhoto(ii), residualsDepth(ii) of the error pass
        const unsigned* row = Ls.slots + (size_t)ti * kPinShort;
        int prev = -1;                               // the source index visited last
        // a long list: the box of the late arrivals, widened by the early ones in the slot row; scanned row by row, in index order
        int scan_r = 0, scan_c = 0, c_lo = 0, c_hi = -1, hi = -1;
        if (m > kPinShort) {
            for (int k = 0; k < kPinShort; ++k) {
                const int e = (int)row[k], c = e % lv.cols;
                box.x = min(box.x, e); box.y = max(box.y, e);
                box.z = min(box.z, c); box.w = max(box.w, c);
            }
            hi = box.y;
            c_lo = box.z; c_hi = box.w;
            scan_r = box.x / lv.cols;
            scan_c = c_lo;
        }
        for (int visited = 0; visited < m; ++visited) {
            int src;
            if (m <= kPinShort) {                    // the next larger index among the slots
                unsigned best = 0xffffffffu;
                for (int k = 0; k < m; ++k) {
                    const unsigned c = row[k];
                    if ((int)c > prev && c < best) best = c;
                }
                src = (int)best;
            } else {                                 // the next source pixel with this key, in index order (there are m of them in the box)
                for (;;) {
                    const int e = scan_r * lv.cols + scan_c;
                    if (++scan_c > c_hi) { scan_c = c_lo; ++scan_r; }
                    if (e > hi || keys[e] == ti) { src = min(e, hi); break; }      // (e > hi cannot happen while visited < m: a bound, not a path)
                }
            }
            prev = src;
            const unsigned v = vals[src];
            const float4 s = lv.src[v & kPinOccIndex];
            float X, Y, Z, iz;
            bool vis;
            (void)warp_pinhole(T, s.x, s.y, s.z, K, lv.rows, lv.cols, X, Y, Z, iz, vis, lv.libm);
            // the photometric and depth residuals of this pixel on this target (both passes use the same expressions)
            float wpf = 0.f, rp = 0.f, wd = 0.f, rd = 0.f;
            {
#pragma clang fp contract(fast)
                if (METHOD != 1) {
                    const float photoDiff = tp.a - s.w;
                    wpf = weight_huber_fast(photoDiff, ec.sigma_photo) * ec.sigma_photo_inv_f;
                    rp = wpf * photoDiff;
                }
                if (METHOD != 0) {
                    const float depthDiff = depth2 - Z;
                    const float sd = ec.sigma_depth * Z;
                    wd = weight_huber_fast(depthDiff, sd) * fast_rcp(sd);
                    rd = wd * depthDiff;
                }
            }
            if (v & kPinOccErr) {                    // errorPhotoICP_Occ1 / _Occ2
                if (!(buf_e > 0.f && iz < buf_e)) {  // RPI.h:1248-1250
                    buf_e = iz;
                    bool on = true;
                    if (METHOD != 1) {
                        if (!sal_p) on = false;      // `continue`: skips the depth part too (RPI.h:1254-1256)
                        else { res_p = rp * rp; ++nP; }
                    }
                    if (METHOD != 0 && on && fin_d && sal_d) { res_d = rd * rd; ++nD; }      // RPI.h:1280-1296
                }
            }
            if (v & kPinOccHess) {                   // calcHessGrad_Occ1 / _Occ2
                bool acc = true;
                if (buf_h == 0.f) ++nVis;            // RPI.h:1421-1430: the first arrival is counted here and below
                else if (iz < buf_h) acc = false;
                if (acc) {
                    ++nVis;
                    buf_h = iz;
                    // rows exist where the photometric residual was stored and is non-zero (RPI.h:1523, 1531: both sums test it)
                    if (METHOD != 1 && sal_p && rp != 0.f) {
#pragma clang fp contract(fast)
                        const float iz2 = iz * iz;
                        const float wgx = wpf * tp.b * K.fx, wgy = wpf * tp.c * K.fy;
                        accumulate_row(A, wgx * iz, wgy * iz, -(wgx * X + wgy * Y) * iz2, X, Y, Z, rp);
                        if (METHOD == 2 && sal_d && fin_d) {
                            const float gx = td.b * K.fx, gy = td.c * K.fy;
                            accumulate_row(A, wd * (gx * iz), wd * (gy * iz), wd * (-(gx * X + gy * Y) * iz2 - 1.f), X, Y, Z, rd);
                        }
                    }
                }
            }
        }
        A.e2p = res_p;
        A.e2d = res_d;
    }
    // block reduction into one partial row (k_eval_pinhole's layout); the counts are integers: LDS atomics
    __shared__ double red[kPinWalkThreads / 64][kNumPartials];
    __shared__ int cnt[3];
    if (threadIdx.x < 3) cnt[threadIdx.x] = 0;
    __syncthreads();
    if (nP) atomicAdd(&cnt[0], nP);
    if (nD) atomicAdd(&cnt[1], nD);
    if (nVis) atomicAdd(&cnt[2], nVis);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        float v[32], out[2];
#pragma unroll
        for (int k = 0; k < 27; ++k) v[k] = A.acc[k];
        v[P_E2P] = A.e2p;
        v[P_E2D] = A.e2d;
        v[P_NP] = v[P_ND] = v[P_NVIS] = 0.f;
        wave_reduce32(v, out);
        if ((lane & 3) == 0) {
            const int row = lane >> 4, quad = (lane >> 2) & 3;
            const int idx = 2 * (quad & 1) + 4 * (quad >> 1) + 8 * (row & 1) + 16 * (row >> 1);
            if (idx + 0 < P_NP) red[wave][idx + 0] = (double)out[0];
            if (idx + 1 < P_NP) red[wave][idx + 1] = (double)out[1];
        }
    }
    __syncthreads();
    if (threadIdx.x < kNumPartials) {
        double v = 0.0;
        if ((int)threadIdx.x < P_NP) {
#pragma unroll
            for (int w = 0; w < kPinWalkThreads / 64; ++w) v += red[w][threadIdx.x];
        } else {
            const int k = (int)threadIdx.x - P_NP;      // P_NP, P_ND, P_NVIS are the last three slots, in this order
            v = k < 3 ? (double)cnt[k] : 0.0;
        }
        partials[(size_t)blockIdx.x * kNumPartials + threadIdx.x] = v;
    }
}

__global__ void k_warp_indices_pinhole(LevelDev lv, PinK K, Pose16 pose, int32_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lv.n) return;
    const PoseRT T = load_pose(pose.v);
    const float4 s = lv.src[i];
    int r = -1, c = -1;
    if (s.x != kInvalidPoint) {
        float X, Y, Z, iz;
        bool vis;
        const unsigned ti = warp_pinhole(T, s.x, s.y, s.z, K, lv.rows, lv.cols, X, Y, Z, iz, vis, lv.libm);
        if (vis) {
            r = (int)(ti / (unsigned)lv.cols);
            c = (int)ti - r * lv.cols;
        }
    }
    out[2 * i] = r;
    out[2 * i + 1] = c;
}

}  // namespace r360
