// Occlusion-aware passes of RegisterPhotoICP:
//   occlusion 1  errorPhotoICP_sphereOcc1 / calcHessGrad_sphereOcc1   RPI.h:3232-3716
//   occlusion 2  errorPhotoICP_sphereOcc2 / calcHessGrad_sphereOcc2   RPI.h:3720-4249
//
// The reference loops run under `#pragma omp parallel for` while reading and writing shared z-buffers and per-target
// rows without synchronisation, so its OpenMP build is timing dependent.  These kernels implement the SEQUENTIAL
// semantics of the same source (pixels visited in index order), which is a function of the inputs only:
//
//   Occ1 error   residual of target pixel t = that of the closest source pixel landing on t (ties: highest index);
//                the valid counters count every source pixel that was the closest *so far* in index order
//                ("prefix maxima" of 1/dist).  H,g: the z-buffer is indexed by the source pixel and never rejects
//                anything; a pixel whose depth gradient is not salient loses its photometric row too.
//   Occ2 error   depth-outlier gate |Dtrg - dist| <= 0.3 m; every prefix maximum contributes its residual (no
//                retraction); both averages divide by the number of prefix maxima.  H,g: per target pixel the rows of
//                the gated source pixel with the HIGHEST index (last writer); numVisible = distinct target pixels.
//
// Implementation: k_occ_build links the candidates of every target pixel into a list (atomicExch on a head array);
// k_occ_resolve (round 2) walks the (short) list of every candidate's target pixel and leaves the three decisions "prefix maximum
// / closest / last" as a flag byte per source pixel; k_eval_occ reads the byte and accumulates the same 32 partial sums per block
// as k_eval, so k_solve is shared.  dist = sqrt_rn(d2) and 1/dist = rcp_rn(dist) are correctly rounded: every comparison is
// bit-for-bit the oracle's.
// Round 1 walked the lists inside k_eval_occ: four to five DEPENDENT global round trips per pixel step (source record -> head ->
// list element -> target records) in a kernel that keeps one 1024-thread block per CU -- 215-300 us per level-0 pass against 15 us
// for the plain pass.  The walk now runs one thread per source pixel in 256-thread blocks (latency hidden by occupancy: 54 us, the
// pass itself 19 us), and the head array needs no memset between passes: its entries carry the pass's generation in their top 8
// bits.  (Measured and dropped: four pixels per thread with their loads issued side by side -- 0.80 ms per alignment with one
// common walk loop, 0.93 ms with only the single-candidate case batched -- instead of 0.64 ms: fewer waves hide less latency than
// the batching gains.)
#pragma once
#include "photo_icp_kernels.h"

namespace r360 {

constexpr float kThresDepthOutliers = 0.3f;      // RPI.h:4525
constexpr int   kOccNotCandidate = -2;

// head entries: generation (top 8 bits) | source pixel index (24 bits; images are < 16 Mpx); an entry of another generation is
// an empty list
__device__ __forceinline__ int occ_decode(int tagged, int gen) { return ((unsigned)tagged >> 24) == (unsigned)gen ? (tagged & 0xFFFFFF) : -1; }

template <int OCC>
__global__ __launch_bounds__(256) void k_occ_build(LevelDev lv, const GNState* __restrict__ st, int level, int gen, int* __restrict__ head,
                                                   int* __restrict__ next, float* __restrict__ dinv, int* __restrict__ tgt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lv.n) return;
    if (st->done || st->level_active != level) return;
    const PoseRT T = load_pose(st->cand);
    const WarpConsts wc = {T.tx, T.ty, T.tz, lv.half_nRows, lv.pi_k};
    const float4 s = lv.src[i];
    float X, Y, Z, rho2, d2;
    bool vis;
    const unsigned ti = warp_pixel(T, wc, s.x, s.y, s.z, lv, X, Y, Z, rho2, d2, vis);
    bool cand = vis && (s.x != kInvalidPoint);
    const float dist = sqrt_rn(d2);
    if (OCC == 2 && cand) {
        const float depth2 = lv.trgD[ti].a;
        if (fabsf(depth2 - dist) > kThresDepthOutliers) cand = false;          // RPI.h:3788-3791, 3968-3979
    }
    int nx = kOccNotCandidate;
    if (cand) {
        dinv[i] = rcp_rn(dist);
        tgt[i] = (int)ti;
        nx = occ_decode(atomicExch(&head[ti], (gen << 24) | i), gen);
    }
    next[i] = nx;
}

// flag byte per source pixel: 1 candidate, 2 prefix maximum, 4 final owner of the z-buffer cell, 8 last writer
__global__ __launch_bounds__(256) void k_occ_resolve(int n, const GNState* __restrict__ st, int level, int gen, const int* __restrict__ head,
                                                     const int* __restrict__ next, const float* __restrict__ dinv,
                                                     const int* __restrict__ tgt, unsigned char* __restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (st->done || st->level_active != level) return;
    unsigned f = 0;
    if (next[i] != kOccNotCandidate) {
        bool pm = true, best = true, last = true;
        const float di = dinv[i];
        for (int j = occ_decode(head[tgt[i]], gen); j >= 0; j = next[j]) {
            if (j == i) continue;
            const float dj = dinv[j];
            if (j < i && dj > di) pm = false;                            // an earlier pixel was closer: occluded on arrival
            if (dj > di || (dj == di && j > i)) best = false;            // not the final owner of the z-buffer cell
            if (j > i) last = false;
        }
        f = 1u | (pm ? 2u : 0u) | (best ? 4u : 0u) | (last ? 8u : 0u);
    }
    flags[i] = (unsigned char)f;
}

template <int METHOD, int OCC>
__global__ __launch_bounds__(kEvalThreads) void k_eval_occ(LevelDev lv, EvalConsts ec, const GNState* __restrict__ st,
                                                            double* __restrict__ partials, int chunk, int level,
                                                            const unsigned char* __restrict__ flags) {
    const int b = blockIdx.x;
    const int base = b * chunk;
    const int end = min(base + chunk, lv.n);
    if (st->done || st->level_active != level) return;
    const PoseRT T = load_pose(st->cand);
    const WarpConsts wc = make_warp_consts(T, lv);

    EvalAcc A;
#pragma unroll
    for (int k = 0; k < 27; ++k) A.acc[k] = 0.f;
    A.e2p = A.e2d = 0.f;
    A.nP = A.nD = A.nVis = 0;

    const int n_steps = (end - base + kEvalThreads - 1) / kEvalThreads;      // wave-uniform: the ballots count whole waves
    for (int k = 0; k < n_steps; ++k) {
        const int i = base + k * kEvalThreads + (int)threadIdx.x;
        const bool in_range = i < end;
        const int ic = in_range ? i : lv.n - 1;
        const float4 s = lv.src[ic];
        float X, Y, Z, rho2, d2;
        bool vis;
        unsigned ti = warp_pixel(T, wc, s.x, s.y, s.z, lv, X, Y, Z, rho2, d2, vis);
        const unsigned fl = in_range ? (unsigned)flags[ic] : 0u;             // k_occ_resolve's decisions at this pose
        const bool cand = (fl & 1u) != 0, pm = (fl & 2u) != 0, best = (fl & 4u) != 0, last = (fl & 8u) != 0;
        ti = cand ? ti : 0u;
        const bool err_on = OCC == 1 ? best : pm;       // whose residual is in the sum
        const bool hg_on = OCC == 1 ? cand : last;      // whose rows reach the normal equations
        A.nVis += ballot_count(hg_on);

        const float dist = sqrt_rn(d2);
        const float dist_inv = rcp_rn(dist);
        float a1, a2, b0, b1, b2;
        {
#pragma clang fp contract(fast)
            const float inv_rho = fast_rsq(rho2);
            const float k_rho2 = lv.angle_res_inv * (inv_rho * inv_rho);
            a1 = k_rho2 * Z;
            a2 = -k_rho2 * Y;
            const float k_d2 = lv.angle_res_inv * (dist_inv * dist_inv);
            b0 = -k_d2 * (rho2 * inv_rho);
            const float c = k_d2 * inv_rho * X;
            b1 = c * Y;
            b2 = c * Z;
        }
        F3 tp = {0.f, 0.f, 0.f}, td = {0.f, 0.f, 0.f};
        if (METHOD != 1) tp = lv.trgP[ti];
        if (METHOD != 0) td = lv.trgD[ti];
        const float depth2 = td.a;
        const bool nonsal_p = METHOD != 1 && fabsf(tp.b) < ec.thr_photo && fabsf(tp.c) < ec.thr_photo;
        const bool nonsal_d = METHOD != 0 && fabsf(td.b) < ec.thr_depth && fabsf(td.c) < ec.thr_depth;
        const bool depth_ok = METHOD != 0 && !nonsal_p && isfinite(depth2) && !nonsal_d;

        if (OCC == 2) {                                  // nValidDepthPts: counted before any saliency test, divides both sums
            const int c = ballot_count(pm);
            A.nP += c;
            A.nD += c;
        }
        if (METHOD != 1) {
            if (OCC == 1) A.nP += ballot_count(pm && !nonsal_p);
            // rows are stored behind the depth block: its `continue` drops the photometric row as well
            const bool row_on = hg_on && !nonsal_p && (METHOD == 0 || !isfinite(depth2) || !nonsal_d);
            if ((err_on || row_on) && !nonsal_p) {
#pragma clang fp contract(fast)
                const float photoDiff = tp.a - s.w;
                const float wpf = weight_huber_fast(photoDiff, ec.sigma_photo) * ec.sigma_photo_inv_f;
                const float res = wpf * photoDiff;
                if (err_on) A.e2p += res * res;
                if (row_on) {
                    const float wgx = wpf * tp.b, wgy = wpf * tp.c;
                    accumulate_row(A, wgy * b0, wgx * a1 + wgy * b1, wgx * a2 + wgy * b2, X, Y, Z, res);
                }
            }
        }
        if (METHOD != 0) {
            if (OCC == 1) A.nD += ballot_count(pm && depth_ok);
            const bool row_on = hg_on && depth_ok;
            if ((err_on || row_on) && depth_ok) {
#pragma clang fp contract(fast)
                const float depthDiff = depth2 - dist;
                const float sd = ec.sigma_depth * depth2;
                const float wd = weight_huber_fast(depthDiff, sd) * fast_rcp(sd);
                const float res = wd * depthDiff;
                if (err_on) A.e2d += res * res;
                if (row_on) {
                    const float kx = wd * (td.c * b0 - X * dist_inv);
                    const float ky = wd * ((td.b * a1 + td.c * b1) - Y * dist_inv);
                    const float kz = wd * ((td.b * a2 + td.c * b2) - Z * dist_inv);
                    accumulate_row(A, kx, ky, kz, X, Y, Z, res);
                }
            }
        }
    }

    // ---- reduction: same partial-row layout as k_eval ----
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

}  // namespace r360
