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
// Implementation (round 2, second form).  Candidates of one target pixel are linked into a list (atomicExch on a generation-tagged
// head array) -- but a list node is a RUN: consecutive source pixels that land on the same target pixel, summarised by the run's
// last lane as {max 1/dist and the pixel that has it, first pixel, last pixel = the node's own index}.  k_occ_build (one thread per
// source pixel, a wave = 64 consecutive pixels) finds the runs with lane shifts, takes a segmented max-scan of (1/dist, index)
// over each run -- which also tells every member whether an EARLIER member of its run is closer -- and leaves a byte per source
// pixel {candidate, prefix maximum within its run, offset to the run's first pixel}.  k_eval_occ, which warps its source pixel
// anyway, reads that byte, gathers its target's list head together with the target records, and decides "prefix maximum / closest /
// last" against the run nodes: earlier runs (first pixel smaller) that hold something closer, any run with a larger (1/dist, index)
// key, any run that ends later.  In the common case -- the pixel is a run of one and alone on its target -- the head word says so
// by itself (no "more than one run" bit, see occ_decode) and no node is read at all.  dist = sqrt_rn(d2) and 1/dist =
// rcp_rn(dist) are correctly rounded and identical in both kernels: every comparison is bit-for-bit the oracle's, and nothing
// depends on the order in which the atomics arrive.
// Why runs: the long lists are at the poles of the sphere, where a hundred and more pixels of a few neighbouring rows collapse
// onto one target pixel (2048 x 1024: up to ~240); every design that walks per-pixel lists pays that length as a chain of
// dependent global loads in some thread: round 1 inside k_eval_occ (one 1024-thread block per CU): 215-300 us per level-0 pass;
// round 2's first form in a kernel of its own (k_occ_resolve, a flag byte per source pixel): 54 us + 19 us for the pass; an attempt
// with two inline slots per target and per-pixel overflow lists walked by k_eval_occ: 170-280 us again.  Those pixels arrive as a
// few long runs, so the run lists are a handful of nodes.
#pragma once
#include "photo_icp_kernels.h"

namespace r360 {

constexpr float kThresDepthOutliers = 0.3f;      // RPI.h:4525

// head entries: "more than one run" (bit 31) | generation (7 bits) | source pixel index (24 bits; images are < 16 Mpx); an entry of
// another generation is an empty list.  Bit 31 is set by every run that finds the list non-empty when it links itself in (an atomic
// OR behind its exchange: the last exchange on a list with two or more runs is always followed by one), so a pass that finds its own
// pixel at the head of a list WITHOUT the bit knows that list is its own node and nothing else -- the common case by far -- and reads
// no node at all (round 4; until then every pixel loaded its own 16-byte node speculatively: 61 -> 45 B per pixel).
constexpr int kOccGenMax = 127;
constexpr unsigned kOccMulti = 0x80000000u;
__device__ __forceinline__ int occ_decode(int tagged, int gen) { return (((unsigned)tagged >> 24) & 0x7Fu) == (unsigned)gen ? (tagged & 0xFFFFFF) : -1; }

// Wave scans on the VALU's data-parallel-primitive paths (row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, then lane 15 of a row to the
// next row and lane 31 to the upper half) instead of __shfl_up steps through the LDS crossbar: k_occ_build made 27 crossbar trips per wave,
// 3456 per CU and launch at 2048 x 1024, on the one LDS unit its four SIMDs share.
#define R360_DPP(old_, src_, ctrl_, rows_) __builtin_amdgcn_update_dpp((int)(old_), (int)(src_), ctrl_, rows_, 0xF, false)
__device__ __forceinline__ int occ_scan_max(int x) {                 // inclusive max-scan of values >= 0
    int t;
    t = R360_DPP(-1, x, 0x111, 0xF); x = t > x ? t : x;
    t = R360_DPP(-1, x, 0x112, 0xF); x = t > x ? t : x;
    t = R360_DPP(-1, x, 0x114, 0xF); x = t > x ? t : x;
    t = R360_DPP(-1, x, 0x118, 0xF); x = t > x ? t : x;
    t = R360_DPP(-1, x, 0x142, 0xA); x = t > x ? t : x;
    t = R360_DPP(-1, x, 0x143, 0xC); x = t > x ? t : x;
    return x;
}
// segmented inclusive max-scan of a 64-bit key over the runs `lead` names (lead = lane of the run's first member, constant along a run,
// increasing from run to run): a source lane counts when it has the same lead (lanes without a source offer lead -1)
__device__ __forceinline__ unsigned long long occ_seg_scan_max(unsigned long long key, int lead) {
    unsigned lo = (unsigned)key, hi = (unsigned)(key >> 32);
#define R360_SEG_STEP(ctrl_, rows_)                                                                              \
    {                                                                                                            \
        const int ol = R360_DPP(-1, lead, ctrl_, rows_);                                                         \
        const unsigned olo = (unsigned)R360_DPP(0, lo, ctrl_, rows_), ohi = (unsigned)R360_DPP(0, hi, ctrl_, rows_); \
        const bool take = ol == lead && (ohi > hi || (ohi == hi && olo > lo));                                   \
        lo = take ? olo : lo;                                                                                    \
        hi = take ? ohi : hi;                                                                                    \
    }
    R360_SEG_STEP(0x111, 0xF)
    R360_SEG_STEP(0x112, 0xF)
    R360_SEG_STEP(0x114, 0xF)
    R360_SEG_STEP(0x118, 0xF)
    R360_SEG_STEP(0x142, 0xA)
    R360_SEG_STEP(0x143, 0xC)
#undef R360_SEG_STEP
    return ((unsigned long long)hi << 32) | lo;
}

// runinfo byte of a source pixel: bit 6 candidate, bit 7 no earlier member of its run is closer, bits 0-5 offset to the run's first pixel
// node (int4, indexed by the run's LAST pixel): x = pixel holding the run's largest (1/dist, index) key, y = bits of that 1/dist,
// z = first pixel of the run, w = next node of the same target pixel (-1: none)
// the node a run's last pixel writes once the exchange on its target's list head has come back (the only thing that waits for it)
struct OccNodePending {
    bool on;
    int i, first, old_head;
    unsigned ti;
    unsigned long long key;
};
__device__ __forceinline__ void occ_store_node(const OccNodePending& p, const int gen, int4* __restrict__ nodes, int* __restrict__ head) {
    if (p.on) {
        const int nx = occ_decode(p.old_head, gen);
        nodes[p.i] = make_int4((int)(unsigned)p.key, (int)(unsigned)(p.key >> 32), p.first, nx);
        if (nx >= 0) atomicOr(&head[p.ti], (int)kOccMulti);      // the list held a run already
    }
}

// one source pixel of the build (every lane of the wave takes part in the shuffles; `in` = the lane has a pixel, ic = its clamped index)
template <int OCC>
__device__ __forceinline__ void occ_build_px(const LevelDev& lv, const PoseRT& T, const WarpConsts& wc, const int i, const bool in, const int ic,
                                             const float4 s, const int lane, const int gen, int* __restrict__ head,
                                             unsigned char* __restrict__ runinfo, OccNodePending& out) {
    float X, Y, Z, rho2, d2;
    bool vis;
    const unsigned ti = warp_pixel(T, wc, s.x, s.y, s.z, lv, X, Y, Z, rho2, d2, vis);
    bool cand = in && vis && (s.x != kInvalidPoint);
    const float dist = sqrt_rn(d2);
    if (OCC == 2 && cand) {
        const float depth2 = lv.trgD[ti].a;
        if (fabsf(depth2 - dist) > kThresDepthOutliers) cand = false;          // RPI.h:3788-3791, 3968-3979
    }
    const float di = rcp_rn(dist);
    // runs of consecutive lanes with the same target pixel (a non-candidate matches nobody: targets are < 2^24)
    const unsigned tkey = cand ? ti : (0xFF000000u | (unsigned)lane);
    // neighbours' keys by wave shifts (lane 0 / lane 63 take a key nobody has)
    const unsigned t_before = (unsigned)R360_DPP(0xFE000000u, tkey, 0x138, 0xF), t_after = (unsigned)R360_DPP(0xFE000000u, tkey, 0x130, 0xF);
    const bool run_head = lane == 0 || t_before != tkey;
    const bool run_tail = lane == 63 || t_after != tkey;
    int lead = lane;
    unsigned long long key = ((unsigned long long)__float_as_uint(di) << 32) | (unsigned)ic;
    bool pm_run = true;
    // Away from the poles and from depth edges no two neighbouring source pixels land on one target pixel: every lane of the wave is a
    // run of its own, and the scans (a third of the kernel's vector instructions) would return what each lane already holds.
    if (__builtin_amdgcn_ballot_w64(!run_head) != 0ull) {      // uniform
        lead = occ_scan_max(run_head ? lane : 0);            // lane of the run's first member
        // segmented inclusive max-scan of the key (1/dist bits, pixel): positive floats order like their bit patterns
        key = occ_seg_scan_max(key, lead);
        // the run's maximum BEFORE this member
        const unsigned long long kprev = ((unsigned long long)(unsigned)R360_DPP(0, (unsigned)(key >> 32), 0x138, 0xF) << 32) | (unsigned)R360_DPP(0, (unsigned)key, 0x138, 0xF);
        const bool has_prev = lane > lead;
        pm_run = !(has_prev && __uint_as_float((unsigned)(kprev >> 32)) > di);
    }
    if (in) runinfo[i] = cand ? (unsigned char)(0x40u | (pm_run ? 0x80u : 0u) | (unsigned)(lane - lead)) : (unsigned char)0;
    out.on = cand && run_tail;
    out.i = i;
    out.key = key;
    out.first = i - (lane - lead);
    out.old_head = 0;
    out.ti = ti;
    if (out.on) out.old_head = atomicExch(&head[ti], (int)(((unsigned)gen << 24) | (unsigned)i));      // (gen <= kOccGenMax)
}

template <int OCC>
__global__ __launch_bounds__(256) void k_occ_build(LevelDev lv, const GNState* __restrict__ st, int level, int gen, int* __restrict__ head,
                                                   int4* __restrict__ nodes, unsigned char* __restrict__ runinfo) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (st->done || st->level_active != level) return;
    const bool in = i < lv.n;                          // (no early exit: every lane takes part in the shuffles)
    const int ic = in ? i : lv.n - 1;
    const PoseRT T = load_pose(st->cand);
    const WarpConsts wc = {T.tx, T.ty, T.tz, lv.half_nRows, lv.pi_k};
    OccNodePending nd;
    occ_build_px<OCC>(lv, T, wc, i, in, ic, lv.src[ic], lane, gen, head, runinfo, nd);
    occ_store_node(nd, gen, nodes, head);
}

// The occlusion-aware schedule with the solve fused in (round 4): k_occ_build_fs is the build on the pass's own grid (blocks of 1024
// threads that walk a span of `chunk` pixels: a wave still owns 64 consecutive pixels per step, chunk is a multiple of 1024, so the
// runs -- and every byte the build leaves -- are k_occ_build's) behind the prologue of k_eval_fs: every block first solves the pass the
// previous k_eval_occ left pending (cfg.occ selects the occlusion modes' error, RPI.h:3232-4249), block 0 writes the new state into
// st_out, and k_eval_occ -- the next launch on the stream -- reads its gate and its pose from there.  An iteration is two launches
// (build + pass) instead of three (build, pass, k_solve).
template <int OCC>
__global__ __launch_bounds__(kEvalThreads) void k_occ_build_fs(const GNState* __restrict__ st_in, GNState* __restrict__ st_out,
                                                                const double* __restrict__ partials_in, int chunk, int level, int nb,
                                                                int pend_rows_hint, LevelDev lv, SolveCfg cfg, FsInit init, int gen,
                                                                int* __restrict__ head, int4* __restrict__ nodes,
                                                                unsigned char* __restrict__ runinfo) {
    __shared__ SolveShared sh;
#ifdef RGBD360_SOLVE_STAMPS
    if (threadIdx.x == 0) sh.stamp0 = __builtin_amdgcn_s_memrealtime();
#endif
    int pend = stage_pending(sh, st_in, partials_in, pend_rows_hint);
    if (init.on) {                              // uniform: the first launch of a schedule initialises the state (k_eval_fs does the same)
        if (threadIdx.x == 0) level_init_one(&sh.sst, init.pose, 1, 1, level);
        __syncthreads();
        pend = 0;
    }
    SOLVE_STAMP(0);
    const int base = blockIdx.x * chunk;
    const int end = min(base + chunk, lv.n);
    const int lane = threadIdx.x & 63;
    auto index_of = [&](int k, bool& in_range) {
        const int i = base + k * kEvalThreads + (int)threadIdx.x;
        in_range = i < end;
        return in_range ? i : lv.n - 1;
    };
    // the first source records, which depend on nothing, are in flight during the solve
    bool in_a, in_b;
    const int ic_a = index_of(0, in_a), ic_b = index_of(1, in_b);
    float4 s_a = lv.src[ic_a], s_b = lv.src[ic_b];
    PoseRT T;
    const bool run = fs_solve_and_publish(sh, pend, cfg, level, nb, lv.n, st_out, T);
    if (!run) return;
    const WarpConsts wc = {T.tx, T.ty, T.tz, lv.half_nRows, lv.pi_k};
    const int n_steps = (end - base + kEvalThreads - 1) / kEvalThreads;      // uniform
    // a step's node is stored behind the arithmetic of the NEXT step: the exchange's round trip is covered, not waited for
    OccNodePending prev = {false, 0, 0, 0, 0u, 0ull};
    for (int k = 0; k < n_steps; ++k) {
        bool in;
        const int ic = index_of(k, in);
        const float4 s = s_a;
        s_a = s_b;
        if (k + 2 < n_steps) {                  // uniform
            bool in_c;
            s_b = lv.src[index_of(k + 2, in_c)];
        }
        OccNodePending cur;
        occ_build_px<OCC>(lv, T, wc, base + k * kEvalThreads + (int)threadIdx.x, in, ic, s, lane, gen, head, runinfo, cur);
        occ_store_node(prev, gen, nodes, head);
        prev = cur;
    }
    occ_store_node(prev, gen, nodes, head);
}

template <int METHOD, int OCC>
__global__ __launch_bounds__(kEvalThreads) void k_eval_occ(LevelDev lv, EvalConsts ec, const GNState* __restrict__ st,
                                                            double* __restrict__ partials, int chunk, int level, int gen,
                                                            const int* __restrict__ head, const int4* __restrict__ nodes,
                                                            const unsigned char* __restrict__ runinfo) {
    const int b = blockIdx.x;
    const int base = b * chunk;
    const int end = min(base + chunk, lv.n);
    EvalAcc A;
#pragma unroll
    for (int k = 0; k < 27; ++k) A.acc[k] = 0.f;
    A.e2p = A.e2d = 0.f;
    A.nP = A.nD = A.nVis = 0;

    const int n_steps = (end - base + kEvalThreads - 1) / kEvalThreads;      // wave-uniform: the ballots count whole waves
    // Two stages, software-pipelined like the plain pass (round 4: the pass is a chain of memory round trips, not issue-bound): while
    // the arithmetic of step k runs, the gathers of step k + 1 (list head + target records at its warped pixel) and what step k + 2
    // needs that does not depend on its warp (source record, run byte) are in flight -- a step waits for neither.
    auto index_of = [&](int k, bool& in_range) {
        const int i = base + k * kEvalThreads + (int)threadIdx.x;
        in_range = i < end;
        return in_range ? i : lv.n - 1;
    };
    struct Pre { float4 s; unsigned info; int ic; bool in; };
    auto preload = [&](int k) {
        Pre p;
        p.ic = index_of(k, p.in);
        p.s = lv.src[p.ic];
        p.info = (unsigned)runinfo[p.ic];
        return p;
    };
    // the first source records and run bytes do not depend on the pose: requested before the state (gate, pose) is read, so that their
    // round trip and the state's overlap (a coarse-level pass is one step: a chain of round trips and nothing else)
    Pre pre = preload(0);
    if (st->done || st->level_active != level) return;
    const PoseRT T = load_pose(st->cand);
    const WarpConsts wc = make_warp_consts(T, lv);
    struct Stage { float X, Y, Z, rho2, d2, sw; unsigned info; int ic, hd_raw; F3 tp, td; };
    auto warp_issue = [&](const Pre& p) {
        Stage w;
        bool vis;
        unsigned ti = warp_pixel(T, wc, p.s.x, p.s.y, p.s.z, lv, w.X, w.Y, w.Z, w.rho2, w.d2, vis);
        w.info = p.in ? p.info : 0u;                                         // k_occ_build's run record at this pose
        ti = (w.info & 0x40u) != 0 ? ti : 0u;
        w.hd_raw = head[ti];
        w.tp = {0.f, 0.f, 0.f};
        w.td = {0.f, 0.f, 0.f};
        if (METHOD != 1) w.tp = lv.trgP[ti];
        if (METHOD != 0) w.td = lv.trgD[ti];
        w.sw = p.s.w;
        w.ic = p.ic;
        return w;
    };
    Stage nxt = warp_issue(pre);
    if (n_steps > 1) pre = preload(1);
    for (int k = 0; k < n_steps; ++k) {
        const Stage w = nxt;
        if (k + 1 < n_steps) {                                               // uniform
            nxt = warp_issue(pre);
            if (k + 2 < n_steps) pre = preload(k + 2);
        }
        const float X = w.X, Y = w.Y, Z = w.Z, rho2 = w.rho2, d2 = w.d2;
        const unsigned info = w.info;
        const int ic = w.ic;
        const F3 tp = w.tp, td = w.td;
        struct { float w; } s = {w.sw};
        const bool cand = (info & 0x40u) != 0;
        const int hd = occ_decode(w.hd_raw, gen);
        const float dist = sqrt_rn(d2);
        const float dist_inv = rcp_rn(dist);
        // the three decisions against the run nodes of the same target pixel
        bool pm = cand && (info & 0x80u) != 0, best = cand, last = cand;
        {
            const unsigned long long key_i = ((unsigned long long)__float_as_uint(dist_inv) << 32) | (unsigned)ic;
            const int my_first = ic - (int)(info & 63u);
            int node = cand ? hd : -1;
            // the head is this pixel, no second run ever linked itself in, and the pixel starts its run: the list is its own node, a
            // run of one (a head is the LAST pixel of its run) -- nothing to compare with, nothing to load
            if (w.hd_raw >= 0 && node == ic && (info & 63u) == 0u) node = -1;
            while (node >= 0) {
                const int4 nd = nodes[node];
                const unsigned long long k = ((unsigned long long)(unsigned)nd.y << 32) | (unsigned)nd.x;
                if (nd.z < my_first && __uint_as_float((unsigned)nd.y) > dist_inv) pm = false;      // an earlier run holds something closer
                if (k > key_i) best = false;                                                        // not the final owner of the z-buffer cell
                if (node > ic) last = false;                                                        // a run ends later
                node = nd.w;
            }
        }
        const bool err_on = OCC == 1 ? best : pm;       // whose residual is in the sum
        const bool hg_on = OCC == 1 ? cand : last;      // whose rows reach the normal equations
        A.nVis += ballot_count(hg_on);

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
