// rgbd360_api.hip -- host side of the C ABI declared in include/rgbd360_hip.h.
//
// Mirrors the call protocol of RegisterPhotoICP (RPI.h:201-288 setters, 480-516 frame setup, 4519-4784
// alignFrames360): frames are converted to device-resident pyramids once; an alignment is a sequence of
// {fused per-pixel pass, solve} launches per pyramid level on one HIP stream, with the Gauss-Newton state
// (poses, H, g, error, iteration counters) living in device memory.  The host only polls a "level done" flag.
// There is no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <thread>
#include <vector>


#include <memory>

#include "../../include/rgbd360_hip.h"
#include "../../include/rgbd360_hip_diag.h"
#include "knobs.h"
#include "host_wait.h"
#include "photo_icp_kernels.h"
#include "occlusion_kernels.h"
#include "pinhole_kernels.h"
#include "f360_state.h"

using namespace r360;

namespace {

struct Level {
    int rows = 0, cols = 0, n = 0;
    float half_nRows = 0.f, angle_res_inv = 0.f;
    float *graySrc = nullptr, *depthSrc = nullptr, *grayTrg = nullptr, *depthTrg = nullptr;
    float4* srcRec = nullptr;
    float4* srcRecPin = nullptr;     // pinhole LUT record of the source (built per alignment, RPI.h:4277-4300)
    F3 *trgP = nullptr, *trgD = nullptr;
    float *sinT = nullptr, *cosT = nullptr, *sinP = nullptr, *cosP = nullptr;
    float2 *tabT = nullptr, *tabP = nullptr;      // the same values interleaved {sin, cos}: one 8-byte load per pixel in the recompute form of the pass
    int nblocks = 0, chunk = 0;
    int libm = 0;                    // rgbd360_set_index_arithmetic: the warp in the reference's libm arithmetic
};

}  // namespace

namespace {
struct SeqEngine;      // sequence_engine.h
}

struct rgbd360_ctx {
    rgbd360_params p;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<Level> levels;
    int rows = 0, cols = 0;
    bool have_src = false, have_trg = false;
    GNState* d_state = nullptr;       // the CURRENT state buffer: every launch of the stream reads / updates this one ...
    GNState* d_state_alt = nullptr;   // ... except the fused pass (k_eval_fs), which reads d_state, writes d_state_alt, after which the two swap
    GNState* h_state = nullptr;   // pinned
    hostwait::SpinTag tag;        // pinned sequence number the stream's last kernel stores (host_wait.h)
    int pend_rows_hint = r360::kPendingRows;      // upper bound of the partial rows the next fused launch finds pending (stage_pending)
    double* d_partials = nullptr;     // partial rows of the last pass enqueued (current) ...
    double* d_partials_alt = nullptr; // ... and where a fused pass puts its rows while its blocks still read the previous table
    int max_blocks = 0;               // rows of a partial table = blocks of the largest level
    bool fused_occ = true;            // the occlusion-aware alignments on the fused schedule too (rgbd360_debug_set_schedule: {build, pass, k_solve} triples)
    bool seq_route_contexts = false;  // rgbd360_debug_set_sequence_route: every sequence over the per-context route (the occlusion-aware ones always are)
    int seq_route_cap = 0;            // ... with this many contexts at most (0: ctx_route_cap())
    bool fused_solve = true;          // single-pair schedule: solve in the prologue of the next pass (rgbd360_debug_set_schedule: {k_eval, k_solve} pairs)
    GnIO* d_gnio = nullptr;
    // upload staging: slot 0 serves the single-frame entries (copies on `stream`); the sequence entry alternates both slots,
    // copying on `up_stream` one frame ahead of the alignment (up_ev: upload landed, conv_ev: slot consumed)
    uint8_t* d_stage_rgb[2] = {nullptr, nullptr};
    uint8_t* d_stage_depth[2] = {nullptr, nullptr};
    size_t stage_rgb_bytes[2] = {0, 0}, stage_depth_bytes[2] = {0, 0};
    hipStream_t up_stream = nullptr;
    hipEvent_t up_ev[2] = {nullptr, nullptr}, conv_ev[2] = {nullptr, nullptr};
    int poll_chunk = 3;           // {pass, solve} pairs per level enqueued ahead of the device
    int first_chunk_top = 8;      // ... and for the first visit of the coarsest level (cheap passes, most iterations)
    int chunk_level0 = 3;         // ... and for the finest level (most expensive passes; a second chunk costs a host round trip)
    bool adaptive_chunks = true;  // RGBD360_ADAPTIVE_CHUNKS=0: fixed chunks only (A/B)
    int hist_iters[8] = {-1, -1, -1, -1, -1, -1, -1, -1};      // accepted iterations per level of the previous alignment: sizes the first chunks of the next
    F360State* f360 = nullptr;        // the Frame360 stages' state (rgbd360_frame360.hip; created on the first Frame360 call, f360_state.h)
    float al_guess[16] = {0};     // alignment in flight (rgbd360_align360_begin / _finish)
    int al_method = 0;
    bool al_active = false;
    std::vector<rgbd360_ctx*> siblings;                // extra contexts of rgbd360_align360_batch's per-context route (owned)
    std::vector<SeqEngine*> engines;                   // lock-step sequence engines of rgbd360_align360_batch (owned)
    int al_occ = 0;
    float cam[4] = {0.f, 0.f, 0.f, 0.f};               // cameraMatrix(0,0), (1,1), (0,2), (1,2)   RPI.h:89, 254-257
    bool have_cam = false;
    float sal_thr = -1.f;                               // useSaliency(true): thresSaliency (RPI.h:217, 266); < 0 = off
    // pinhole occlusion passes: (target index, source index) pairs before / after the sort, the sort's scratch, one partial row per walk block
    unsigned *pin_keys = nullptr, *pin_vals = nullptr;      // the pinhole occlusion passes: per source pixel its target + pass flags,
    PinOccLists pin_lists = {nullptr, nullptr, nullptr};      // per target pixel its arrivals (pinhole_kernels.h)
    size_t pin_occ_n = 0;
    double* pin_partials = nullptr;
    int* occ_head = nullptr;                           // occlusion modes: per-target lists of candidate runs (generation-tagged heads)
    int4* occ_nodes = nullptr;                         // ... run nodes, indexed by the run's last source pixel
    unsigned char* occ_runinfo = nullptr;              // ... per source pixel: candidate / prefix maximum within its run / offset to the run's first pixel
    int occ_gen = 0;                                   // generation tag of the head entries (no memset between passes)
    size_t occ_n = 0;
    int max_eval_blocks = 256;    // grid cap of the fused pass (debug knob RGBD360_EVAL_BLOCKS, csrc/knobs.h)
    int index_libm = 0;           // rgbd360_set_index_arithmetic: 1 = the spherical warp in the reference's libm arithmetic
    unsigned char* arena = nullptr;   // ONE allocation behind every per-level buffer of the context (planes, records, angle tables)
    std::string err;
};

namespace {

#define HIPC(ctx, expr)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                               \
            return -(int)e_ - 1000;                                                                       \
        }                                                                                                 \
    } while (0)

int fail(rgbd360_ctx* ctx, int code, const char* msg) {
    ctx->err = msg;
    return code;
}

void pin_occ_free(rgbd360_ctx* ctx) {
    hipFree(ctx->pin_keys); hipFree(ctx->pin_vals); hipFree(ctx->pin_partials);
    hipFree(ctx->pin_lists.cnt); hipFree(ctx->pin_lists.box); hipFree(ctx->pin_lists.slots);
    ctx->pin_keys = ctx->pin_vals = nullptr;
    ctx->pin_partials = nullptr;
    ctx->pin_lists = PinOccLists{nullptr, nullptr, nullptr};
    ctx->pin_occ_n = 0;
}

void free_levels(rgbd360_ctx* ctx) {
    for (Level& L : ctx->levels) {
        hipFree(L.srcRecPin);
    }
    pin_occ_free(ctx);
    hipFree(ctx->arena);
    ctx->arena = nullptr;
    ctx->levels.clear();
    ctx->rows = ctx->cols = 0;
    ctx->have_src = ctx->have_trg = false;
}

int ensure_levels(rgbd360_ctx* ctx, int rows, int cols) {
    if (ctx->rows == rows && ctx->cols == cols && !ctx->levels.empty()) return 0;
    if (rows < 2 || cols < 8) return fail(ctx, -1, "image too small");
    if ((rows >> (ctx->p.n_pyr - 1)) < 2 || (cols >> (ctx->p.n_pyr - 1)) < 8)
        return fail(ctx, -1, "too many pyramid levels for this image size (coarsest level must be >= 2 x 8)");
    if ((long long)rows * cols >= (1ll << 24) || rows >= (1 << 15) || cols >= (1 << 15))
        return fail(ctx, -1, "image too large (the fused pass uses 24-bit index arithmetic: < 16 Mpx)");
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    free_levels(ctx);
    ctx->levels.resize(ctx->p.n_pyr);
    size_t arena_off = 0;
    {
        size_t total = 0;
        for (int l = 0, rr = rows, cc = cols; l < ctx->p.n_pyr; ++l, rr /= 2, cc /= 2) {
            const size_t n = (size_t)rr * cc;
            total += 4 * ((n * 4 + 255) & ~(size_t)255) + ((n * 16 + 255) & ~(size_t)255) + 2 * ((n * 12 + 255) & ~(size_t)255) +
                     2 * (((size_t)cc * 4 + 255) & ~(size_t)255) + 2 * (((size_t)rr * 4 + 255) & ~(size_t)255) +
                     (((size_t)cc * 8 + 255) & ~(size_t)255) + (((size_t)rr * 8 + 255) & ~(size_t)255);
        }
        HIPC(ctx, hipMalloc(&ctx->arena, total));
    }
    int r = rows, c = cols;
    int max_blocks = 0;
    for (int l = 0; l < ctx->p.n_pyr; ++l) {
        Level& L = ctx->levels[l];
        L.rows = r; L.cols = c; L.n = r * c;
        const float angle_res = 2 * kPI / c;        // RPI.h:2554
        L.angle_res_inv = 1 / angle_res;            // RPI.h:2555
        L.half_nRows = 0.5 * r - 0.5;               // RPI.h:2557
        // every buffer of every level out of ONE allocation (sized below on the first level's visit): a context is ~110 MB at
        // 2048 x 1024, and one large range keeps its pages' translations together (fewer, larger fragments) instead of scattering
        // ~45 small allocations over the address space
        auto take = [&](size_t bytes) -> void* {
            void* q = ctx->arena + arena_off;
            arena_off += (bytes + 255) & ~(size_t)255;
            return q;
        };
        const size_t nb = (size_t)L.n * sizeof(float);
        L.graySrc = (float*)take(nb); L.depthSrc = (float*)take(nb); L.grayTrg = (float*)take(nb); L.depthTrg = (float*)take(nb);
        L.srcRec = (float4*)take((size_t)L.n * sizeof(float4));
        L.trgP = (F3*)take((size_t)L.n * sizeof(F3));
        L.trgD = (F3*)take((size_t)L.n * sizeof(F3));
        L.sinT = (float*)take(c * sizeof(float)); L.cosT = (float*)take(c * sizeof(float));
        L.sinP = (float*)take(r * sizeof(float)); L.cosP = (float*)take(r * sizeof(float));
        L.tabT = (float2*)take(c * sizeof(float2)); L.tabP = (float2*)take(r * sizeof(float2));
        if (!L.tabT || !L.tabP) return fail(ctx, -103, "cannot allocate the pyramid levels");
        if (!L.graySrc || !L.depthSrc || !L.grayTrg || !L.depthTrg || !L.srcRec || !L.trgP || !L.trgD || !L.sinT || !L.cosT || !L.sinP || !L.cosP)
            return fail(ctx, -103, "cannot allocate the pyramid levels");
        // RPI.h:4556-4571: per-column / per-row sin, cos of float arguments (host libm, once per size)
        std::vector<float> st(c), ct(c), sp(r), cp(r);
        for (int j = 0; j < c; ++j) {
            float theta = j * angle_res;
            st[j] = sinf(theta);
            ct[j] = cosf(theta);
        }
        for (int i = 0; i < r; ++i) {
            float phi = (L.half_nRows - i) * angle_res;
            sp[i] = sinf(phi);
            cp[i] = cosf(phi);
        }
        HIPC(ctx, hipMemcpy(L.sinT, st.data(), c * sizeof(float), hipMemcpyHostToDevice));
        HIPC(ctx, hipMemcpy(L.cosT, ct.data(), c * sizeof(float), hipMemcpyHostToDevice));
        HIPC(ctx, hipMemcpy(L.sinP, sp.data(), r * sizeof(float), hipMemcpyHostToDevice));
        HIPC(ctx, hipMemcpy(L.cosP, cp.data(), r * sizeof(float), hipMemcpyHostToDevice));
        {
            std::vector<float2> tt(c), tp(r);
            for (int j = 0; j < c; ++j) tt[j] = make_float2(st[j], ct[j]);
            for (int i = 0; i < r; ++i) tp[i] = make_float2(sp[i], cp[i]);
            HIPC(ctx, hipMemcpy(L.tabT, tt.data(), c * sizeof(float2), hipMemcpyHostToDevice));
            HIPC(ctx, hipMemcpy(L.tabP, tp.data(), r * sizeof(float2), hipMemcpyHostToDevice));
        }
        // work split of the fused pass: <= max_eval_blocks blocks, contiguous spans that are multiples of 256 pixels
        int chunk = (L.n + ctx->max_eval_blocks - 1) / ctx->max_eval_blocks;
        chunk = ((chunk + kEvalThreads - 1) / kEvalThreads) * kEvalThreads;
        L.chunk = chunk;
        L.nblocks = (L.n + chunk - 1) / chunk;
        L.libm = ctx->index_libm;
        if (L.nblocks > max_blocks) max_blocks = L.nblocks;
        r /= 2; c /= 2;
    }
    hipFree(ctx->d_partials); hipFree(ctx->d_partials_alt);
    ctx->d_partials = ctx->d_partials_alt = nullptr;
    // at least the kPendingRows rows stage_pending always loads, + diagnostic rows
    const size_t part_bytes = (size_t)(std::max((max_blocks + 31) / 32 * 32, kPendingRows) + 32 + max_blocks / 2 + 2) * kNumPartials * sizeof(double);      // + the diagnostic rows of the stamp builds
    HIPC(ctx, hipMalloc(&ctx->d_partials, part_bytes));
    HIPC(ctx, hipMalloc(&ctx->d_partials_alt, part_bytes));
    HIPC(ctx, hipMemset(ctx->d_partials, 0, part_bytes));          // the fused pass loads max_blocks rows whatever the pending count
    HIPC(ctx, hipMemset(ctx->d_partials_alt, 0, part_bytes));
    ctx->max_blocks = max_blocks;
    ctx->rows = rows; ctx->cols = cols;
    return 0;
}

LevelDev level_dev(const Level& L) {
    LevelDev d;
    d.rows = L.rows; d.cols = L.cols; d.n = L.n;
    d.half_nRows = L.half_nRows; d.angle_res_inv = L.angle_res_inv;
    d.pi_k = (float)(kPI * (double)L.angle_res_inv);
    d.src = L.srcRec; d.trgP = L.trgP; d.trgD = L.trgD;
    d.depth_src = L.depthSrc; d.gray_src = L.graySrc; d.tabT = L.tabT; d.tabP = L.tabP;
    d.libm = L.libm;
    return d;
}
// Levels of this many pixels and more run the per-pixel pass in its recompute form (SRC 1, photo_icp_kernels.h: 8 B per source pixel less;
// a gain wherever the level is fed from HBM, a small loss on the latency-bound small levels).  Default: levels whose photo + depth
// working set (40 B/px) does not fit the 256 MiB Infinity Cache next to anything else, i.e. 4096 x 2048 and up; RGBD360_RECOMPUTE_MIN_PX
// moves the bound (0 = every level, for A/B runs and tests).
int recompute_min_px() {
    static const int v = [] { const char* e = knobs::product("RGBD360_RECOMPUTE_MIN_PX"); return e ? atoi(e) : 4 * 1024 * 1024; }();
    return v;
}

EvalConsts eval_consts(const rgbd360_params& p) {
    EvalConsts e;
    e.sigma_photo = p.sigma_photo; e.sigma_depth = p.sigma_depth;
    e.thr_photo = p.thres_sal_photo; e.thr_depth = p.thres_sal_depth;
    e.sigma_photo_inv_f = 1. / p.sigma_photo;
    e.sigma_photo_inv_d = 1. / p.sigma_photo;
    return e;
}

dim3 grid2d(int rows, int cols, int bx = 256) { return dim3((cols + bx - 1) / bx, rows, 1); }

int occ_ensure(rgbd360_ctx* ctx) {
    const size_t n = ctx->levels.empty() ? 0 : (size_t)ctx->levels[0].n;
    if (ctx->occ_n >= n && n > 0) return 0;
    hipFree(ctx->occ_head); hipFree(ctx->occ_nodes); hipFree(ctx->occ_runinfo);
    ctx->occ_head = nullptr;
    ctx->occ_nodes = nullptr;
    ctx->occ_runinfo = nullptr;
    ctx->occ_n = 0;
    HIPC(ctx, hipMalloc(&ctx->occ_head, n * sizeof(int)));
    HIPC(ctx, hipMalloc(&ctx->occ_nodes, n * sizeof(int4)));
    HIPC(ctx, hipMalloc(&ctx->occ_runinfo, n));
    HIPC(ctx, hipMemsetAsync(ctx->occ_head, 0, n * sizeof(int), ctx->stream));      // generation 0 = empty; the passes count from 1
    ctx->occ_gen = 0;
    ctx->occ_n = n;
    return 0;
}

void launch_eval(rgbd360_ctx* ctx, int level, int method, bool hg, int occ = 0) {
    const Level& L = ctx->levels[level];
    LevelDev lv = level_dev(L);
    lv.min_depth = ctx->p.min_depth; lv.max_depth = ctx->p.max_depth;
    const EvalConsts ec = eval_consts(ctx->p);
    dim3 g(L.nblocks), b(kEvalThreads);
    if (occ != 0) {
        // per-target lists of candidate RUNS at the pose under evaluation, then the occlusion-aware fused pass, which decides per pixel.
        // The head entries carry the pass's generation (1..127) in their top byte: one memset per 127 passes instead of one per pass.
        if (++ctx->occ_gen > kOccGenMax) {
            hipMemsetAsync(ctx->occ_head, 0, ctx->occ_n * sizeof(int), ctx->stream);
            ctx->occ_gen = 1;
        }
        const int gen = ctx->occ_gen;
        const dim3 gb((L.n + 255) / 256), bb(256);
        if (occ == 1) hipLaunchKernelGGL((k_occ_build<1>), gb, bb, 0, ctx->stream, lv, ctx->d_state, level, gen, ctx->occ_head, ctx->occ_nodes, ctx->occ_runinfo);
        else hipLaunchKernelGGL((k_occ_build<2>), gb, bb, 0, ctx->stream, lv, ctx->d_state, level, gen, ctx->occ_head, ctx->occ_nodes, ctx->occ_runinfo);
#define LAUNCH_OCC(M, O) hipLaunchKernelGGL((k_eval_occ<M, O>), g, b, 0, ctx->stream, lv, ec, ctx->d_state, ctx->d_partials, L.chunk, level, gen, ctx->occ_head, (const int4*)ctx->occ_nodes, (const unsigned char*)ctx->occ_runinfo)
        if (occ == 1) {
            if (method == 0) LAUNCH_OCC(0, 1);
            else if (method == 1) LAUNCH_OCC(1, 1);
            else LAUNCH_OCC(2, 1);
        } else {
            if (method == 0) LAUNCH_OCC(0, 2);
            else if (method == 1) LAUNCH_OCC(1, 2);
            else LAUNCH_OCC(2, 2);
        }
#undef LAUNCH_OCC
        return;
    }
#define LAUNCH(M, HG, S) hipLaunchKernelGGL((k_eval<M, HG, S>), g, b, 0, ctx->stream, ctx->d_state, lv.src, lv.n, L.chunk, level, L.nblocks, ctx->d_partials, lv, ec)
    const bool rc = L.n >= recompute_min_px();
    if (hg) {
        if (rc) {
            if (method == 0) LAUNCH(0, true, 1);
            else if (method == 1) LAUNCH(1, true, 1);
            else LAUNCH(2, true, 1);
        } else {
            if (method == 0) LAUNCH(0, true, 0);
            else if (method == 1) LAUNCH(1, true, 0);
            else LAUNCH(2, true, 0);
        }
    } else {
        if (method == 0) LAUNCH(0, false, 0);
        else if (method == 1) LAUNCH(1, false, 0);
        else LAUNCH(2, false, 0);
    }
#undef LAUNCH
}

// publish: the solve also writes the state into ctx->h_state and bumps the context's host tag (hostwait::wait picks it up)
void launch_solve(rgbd360_ctx* ctx, int level, int mode, int forced, int occ = 0, bool publish = false) {
    const Level& L = ctx->levels[level];
    SolveCfg cfg;
    cfg.level = level; cfg.mode = mode; cfg.forced = forced; cfg.max_iters = ctx->p.max_iters; cfg.n_pixels = L.n;
    cfg.occ = occ;
    cfg.tol_residual = ctx->p.tol_residual; cfg.tol_update = ctx->p.tol_update;
    if (publish) {
        cfg.host_state = ctx->h_state;
        cfg.host_tag = ctx->tag.h;
        cfg.host_seq = ++ctx->tag.seq;
    }
    hipLaunchKernelGGL(k_solve, dim3(1), dim3(kSolveThreads), 0, ctx->stream, ctx->d_state, ctx->d_partials, L.nblocks, cfg);
}

// Fused-solve schedule (k_eval_fs, photo_icp_kernels.h): the launch solves the pass the previous launch left pending and runs the
// next pass; state and partial table ping-pong, the host's "current" pointers follow the stream order.
bool fused_ok(const rgbd360_ctx* ctx, int occ) {
    return ctx->fused_solve && (occ == 0 || ctx->fused_occ) && ctx->max_blocks <= kMaxPendingRows;
}
SolveCfg fused_cfg(const rgbd360_ctx* ctx, int forced, int occ = 0) {
    SolveCfg cfg;
    cfg.level = -1; cfg.n_pixels = 0;       // taken from the state (level_active / pend_npix of the pending pass)
    cfg.mode = 0; cfg.forced = forced; cfg.max_iters = ctx->p.max_iters; cfg.occ = occ;
    cfg.tol_residual = ctx->p.tol_residual; cfg.tol_update = ctx->p.tol_update;
    return cfg;
}
// init_pose != nullptr: the launch starts the schedule itself at that pose (no k_level_init in front of it)
void launch_eval_fused(rgbd360_ctx* ctx, int level, int method, int forced, const float* init_pose = nullptr) {
    FsInit init;
    init.on = init_pose ? 1 : 0;
    if (init_pose) memcpy(init.pose.v, init_pose, sizeof(init.pose.v));
    else memset(init.pose.v, 0, sizeof(init.pose.v));
    const Level& L = ctx->levels[level];
    LevelDev lv = level_dev(L);
    lv.min_depth = ctx->p.min_depth; lv.max_depth = ctx->p.max_depth;
    const EvalConsts ec = eval_consts(ctx->p);
    const SolveCfg cfg = fused_cfg(ctx, forced);
    dim3 g(L.nblocks), b(kEvalThreads);
#define LAUNCHF(M, S) hipLaunchKernelGGL((k_eval_fs<M, S>), g, b, 0, ctx->stream, (const GNState*)ctx->d_state, ctx->d_state_alt, (const double*)ctx->d_partials, \
                                      ctx->d_partials_alt, lv.src, lv.n, L.chunk, level, L.nblocks, ctx->pend_rows_hint, lv, ec, cfg, init)
    if (L.n >= recompute_min_px()) {
        if (method == 0) LAUNCHF(0, 1);
        else if (method == 1) LAUNCHF(1, 1);
        else LAUNCHF(2, 1);
    } else {
        if (method == 0) LAUNCHF(0, 0);
        else if (method == 1) LAUNCHF(1, 0);
        else LAUNCHF(2, 0);
    }
#undef LAUNCHF
    ctx->pend_rows_hint = L.nblocks;             // what this launch can leave pending bounds what the next one has to load
    std::swap(ctx->d_state, ctx->d_state_alt);
    std::swap(ctx->d_partials, ctx->d_partials_alt);
}
// The occlusion-aware iteration of the fused schedule: k_occ_build_fs (solve of the pending pass + run lists at the new pose; writes
// the new state) and k_eval_occ (gate and pose from that state; leaves its rows pending) -- two launches instead of three.
void launch_occ_fused(rgbd360_ctx* ctx, int level, int method, int occ, int forced, const float* init_pose = nullptr) {
    FsInit init;
    init.on = init_pose ? 1 : 0;
    if (init_pose) memcpy(init.pose.v, init_pose, sizeof(init.pose.v));
    else memset(init.pose.v, 0, sizeof(init.pose.v));
    const Level& L = ctx->levels[level];
    LevelDev lv = level_dev(L);
    lv.min_depth = ctx->p.min_depth; lv.max_depth = ctx->p.max_depth;
    const EvalConsts ec = eval_consts(ctx->p);
    const SolveCfg cfg = fused_cfg(ctx, forced, occ);
    if (++ctx->occ_gen > kOccGenMax) {
        hipMemsetAsync(ctx->occ_head, 0, ctx->occ_n * sizeof(int), ctx->stream);
        ctx->occ_gen = 1;
    }
    const int gen = ctx->occ_gen;
    dim3 g(L.nblocks), b(kEvalThreads);
#define LAUNCH_BUILD(O) hipLaunchKernelGGL((k_occ_build_fs<O>), g, b, 0, ctx->stream, (const GNState*)ctx->d_state, ctx->d_state_alt, (const double*)ctx->d_partials, \
                                           L.chunk, level, L.nblocks, ctx->pend_rows_hint, lv, cfg, init, gen, ctx->occ_head, ctx->occ_nodes, ctx->occ_runinfo)
    if (occ == 1) LAUNCH_BUILD(1);
    else LAUNCH_BUILD(2);
#undef LAUNCH_BUILD
#define LAUNCH_OCC(M, O) hipLaunchKernelGGL((k_eval_occ<M, O>), g, b, 0, ctx->stream, lv, ec, (const GNState*)ctx->d_state_alt, ctx->d_partials_alt, L.chunk, level, gen, \
                                            ctx->occ_head, (const int4*)ctx->occ_nodes, (const unsigned char*)ctx->occ_runinfo)
    if (occ == 1) {
        if (method == 0) LAUNCH_OCC(0, 1);
        else if (method == 1) LAUNCH_OCC(1, 1);
        else LAUNCH_OCC(2, 1);
    } else {
        if (method == 0) LAUNCH_OCC(0, 2);
        else if (method == 1) LAUNCH_OCC(1, 2);
        else LAUNCH_OCC(2, 2);
    }
#undef LAUNCH_OCC
    ctx->pend_rows_hint = L.nblocks;
    std::swap(ctx->d_state, ctx->d_state_alt);
    std::swap(ctx->d_partials, ctx->d_partials_alt);
}
// the tail of a fused schedule: solves what the last pass left pending (nothing, if that launch was a no-op) and publishes
void launch_solve_pending(rgbd360_ctx* ctx, int forced, bool publish, int occ = 0) {
    SolveCfg cfg = fused_cfg(ctx, forced, occ);
    if (publish) {
        cfg.host_state = ctx->h_state;
        cfg.host_tag = ctx->tag.h;
        cfg.host_seq = ++ctx->tag.seq;
    }
    hipLaunchKernelGGL(k_solve_pending, dim3(1), dim3(kSolveThreads), 0, ctx->stream, ctx->d_state, (const double*)ctx->d_partials, ctx->pend_rows_hint, cfg);
}

void launch_level_init(rgbd360_ctx* ctx, int level, const float* pose, int reset_all) {
    Pose16 P;
    if (pose) memcpy(P.v, pose, sizeof(P.v));
    else memset(P.v, 0, sizeof(P.v));
    hipLaunchKernelGGL(k_level_init, dim3(1), dim3(64), 0, ctx->stream, ctx->d_state, P, pose ? 1 : 0, reset_all, level);
    ctx->pend_rows_hint = kPendingRows;
}

// the device state into ctx->h_state with memcpy + stream synchronise: for callers that read HIP events afterwards (an event the
// runtime has not yet seen complete costs hipEventElapsedTime a blocking wait of its own: +30 us per call measured)
int read_state_sync(rgbd360_ctx* ctx) {
    HIPC(ctx, hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(GNState), hipMemcpyDeviceToHost, ctx->stream));
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
// the device state into ctx->h_state, and wait for it: publishing kernel + host spin (host_wait.h), not memcpy + stream synchronise
int read_state(rgbd360_ctx* ctx) {
    static_assert(sizeof(GNState) % 4 == 0, "published word by word");
    HIPC(ctx, hostwait::publish_and_wait(ctx->tag, ctx->stream, ctx->d_state, ctx->h_state, sizeof(GNState)));
    return 0;
}

int check_args(rgbd360_ctx* ctx, int level, int method) {
    if (!ctx) return -1;
    if (!ctx->have_src || !ctx->have_trg) return fail(ctx, -2, "set_target and set_source must be called first");
    if (level < 0 || level >= (int)ctx->levels.size()) return fail(ctx, -3, "bad pyramid level");
    if (method < 0 || method > 2) return fail(ctx, -4, "bad method");
    return 0;
}

// every gradient record of the target frame (all levels x {intensity, depth}) in one launch
void launch_gradient_recs(rgbd360_ctx* ctx) {
    GradJobs jobs;
    int nb = 0, n = 0;
    for (int l = 0; l < ctx->p.n_pyr; ++l) {
        const Level& L = ctx->levels[l];
        const int seam = ctx->p.mask_seams ? L.cols / 8 : 0;
        for (int k = 0; k < 2; ++k) {
            jobs.src[n] = k == 0 ? L.grayTrg : L.depthTrg;
            jobs.rec[n] = k == 0 ? L.trgP : L.trgD;
            jobs.rows[n] = L.rows; jobs.cols[n] = L.cols; jobs.seam[n] = seam;
            jobs.first_block[n] = nb;
            nb += (L.n + 255) / 256;
            ++n;
        }
    }
    jobs.first_block[n] = nb;
    jobs.n = n;
    hipLaunchKernelGGL(k_gradient_rec_multi, dim3(nb), dim3(256), 0, ctx->stream, jobs);
}

// every source record (LUT point + intensity) of the source frame in one launch
void launch_src_recs(rgbd360_ctx* ctx) {
    SrcJobs jobs;
    int nb = 0, n = 0;
    for (int l = 0; l < ctx->p.n_pyr; ++l) {
        const Level& L = ctx->levels[l];
        jobs.depth[n] = L.depthSrc; jobs.gray[n] = L.graySrc;
        jobs.sin_theta[n] = L.sinT; jobs.cos_theta[n] = L.cosT; jobs.sin_phi[n] = L.sinP; jobs.cos_phi[n] = L.cosP;
        jobs.rec[n] = L.srcRec;
        jobs.rows[n] = L.rows; jobs.cols[n] = L.cols;
        jobs.first_block[n] = nb;
        nb += (L.n + 255) / 256;
        ++n;
    }
    jobs.first_block[n] = nb;
    jobs.n = n;
    jobs.min_depth = ctx->p.min_depth; jobs.max_depth = ctx->p.max_depth;
    hipLaunchKernelGGL(k_src_rec_multi, dim3(nb), dim3(256), 0, ctx->stream, jobs);
}

static int ensure_stage(rgbd360_ctx* ctx, int slot, size_t need_rgb, size_t need_d) {
    if (ctx->stage_rgb_bytes[slot] < need_rgb) {
        hipFree(ctx->d_stage_rgb[slot]);
        ctx->d_stage_rgb[slot] = nullptr; ctx->stage_rgb_bytes[slot] = 0;
        HIPC(ctx, hipMalloc(&ctx->d_stage_rgb[slot], need_rgb));
        ctx->stage_rgb_bytes[slot] = need_rgb;
    }
    if (ctx->stage_depth_bytes[slot] < need_d) {
        hipFree(ctx->d_stage_depth[slot]);
        ctx->d_stage_depth[slot] = nullptr; ctx->stage_depth_bytes[slot] = 0;
        HIPC(ctx, hipMalloc(&ctx->d_stage_depth[slot], need_d));
        ctx->stage_depth_bytes[slot] = need_d;
    }
    return 0;
}

// Sequence path: copy one host frame into staging slot `slot` on the context's upload stream, ahead of its use.
static int upload_stage(rgbd360_ctx* ctx, int slot, const uint8_t* rgb, size_t rgb_step, const void* depth, size_t d_step,
                        int depth_type, int rows, int cols) {
    const size_t dpx = depth_type == 0 ? 2 : 4;
    if (!ctx->up_stream) {
        HIPC(ctx, hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            HIPC(ctx, hipEventCreateWithFlags(&ctx->up_ev[k], hipEventDisableTiming));
            HIPC(ctx, hipEventCreateWithFlags(&ctx->conv_ev[k], hipEventDisableTiming));
        }
    }
    int rc = ensure_stage(ctx, slot, (size_t)rows * cols * 3, (size_t)rows * cols * dpx);
    if (rc) return rc;
    HIPC(ctx, hipStreamWaitEvent(ctx->up_stream, ctx->conv_ev[slot], 0));      // the slot's previous frame has been converted
    HIPC(ctx, hipMemcpy2DAsync(ctx->d_stage_rgb[slot], (size_t)cols * 3, rgb, rgb_step, (size_t)cols * 3, rows,
                               hipMemcpyHostToDevice, ctx->up_stream));
    HIPC(ctx, hipMemcpy2DAsync(ctx->d_stage_depth[slot], (size_t)cols * dpx, depth, d_step, (size_t)cols * dpx, rows,
                               hipMemcpyHostToDevice, ctx->up_stream));
    HIPC(ctx, hipEventRecord(ctx->up_ev[slot], ctx->up_stream));
    return 0;
}

// staged_slot >= 0: the frame was put into that staging slot by upload_stage (rgb / depth are then unused).
int set_frame(rgbd360_ctx* ctx, bool target, const uint8_t* rgb, size_t rgb_step, const void* depth, size_t d_step,
              int depth_type, int rows, int cols, bool on_device, bool wait_for_upload = true, int staged_slot = -1) {
    if (!ctx) return -1;
    if (staged_slot < 0 && (!rgb || !depth)) return fail(ctx, -1, "null image pointer");
    if (depth_type != 0 && depth_type != 1) return fail(ctx, -1, "depth_type must be 0 (u16 mm) or 1 (f32 m)");
    if (ctx->levels.size() && (rows != ctx->rows || cols != ctx->cols) && (target ? ctx->have_src : ctx->have_trg)) {
        // the other frame has different dimensions: both must be set again (the reference would read out of bounds)
        if (target) ctx->have_src = false; else ctx->have_trg = false;
    }
    int rc = ensure_levels(ctx, rows, cols);
    if (rc) return rc;
    const size_t dpx = depth_type == 0 ? 2 : 4;
    const uint8_t* d_rgb = rgb;
    const void* d_depth = depth;
    size_t s_rgb = rgb_step, s_depth = d_step;
    if (staged_slot >= 0) {
        HIPC(ctx, hipStreamWaitEvent(ctx->stream, ctx->up_ev[staged_slot], 0));
        d_rgb = ctx->d_stage_rgb[staged_slot]; d_depth = ctx->d_stage_depth[staged_slot];
        s_rgb = (size_t)cols * 3; s_depth = (size_t)cols * dpx;
    } else if (!on_device) {
        rc = ensure_stage(ctx, 0, (size_t)rows * cols * 3, (size_t)rows * cols * dpx);
        if (rc) return rc;
        HIPC(ctx, hipMemcpy2DAsync(ctx->d_stage_rgb[0], (size_t)cols * 3, rgb, rgb_step, (size_t)cols * 3, rows,
                                   hipMemcpyHostToDevice, ctx->stream));
        HIPC(ctx, hipMemcpy2DAsync(ctx->d_stage_depth[0], (size_t)cols * dpx, depth, d_step, (size_t)cols * dpx, rows,
                                   hipMemcpyHostToDevice, ctx->stream));
        d_rgb = ctx->d_stage_rgb[0]; d_depth = ctx->d_stage_depth[0];
        s_rgb = (size_t)cols * 3; s_depth = (size_t)cols * dpx;
    }
    Level& L0 = ctx->levels[0];
    float* gray0 = target ? L0.grayTrg : L0.graySrc;
    float* dep0 = target ? L0.depthTrg : L0.depthSrc;
    {
        dim3 g = grid2d(rows, (cols + 3) / 4);
        g.z = 2;                                        // colour -> intensity and depth -> metres in one launch
        hipLaunchKernelGGL(k_convert_pair, g, dim3(256), 0, ctx->stream, d_rgb, s_rgb, d_depth, s_depth, depth_type, rows, cols, gray0, dep0);
        if (staged_slot >= 0) HIPC(ctx, hipEventRecord(ctx->conv_ev[staged_slot], ctx->stream));
    }
    for (int l = 1; l < ctx->p.n_pyr; ++l) {
        Level &P = ctx->levels[l - 1], &C = ctx->levels[l];
        dim3 g = grid2d(C.rows, C.cols);
        g.z = 2;                                        // intensity and depth step in one launch
        hipLaunchKernelGGL(k_pyrdown_pair, g, dim3(256), 0, ctx->stream, target ? P.grayTrg : P.graySrc,
                           target ? P.depthTrg : P.depthSrc, P.rows, P.cols, target ? C.grayTrg : C.graySrc,
                           target ? C.depthTrg : C.depthSrc, C.rows, C.cols, ctx->p.min_depth, ctx->p.max_depth);
    }
    if (target) launch_gradient_recs(ctx);
    else launch_src_recs(ctx);
    HIPC(ctx, hipGetLastError());
    // host buffers may be reused by the caller once a set_* call returns; the sequence entry owns them until it returns
    if (!on_device && staged_slot < 0 && wait_for_upload) HIPC(ctx, hipStreamSynchronize(ctx->stream));
    if (target) ctx->have_trg = true; else ctx->have_src = true;
    return 0;
}

}  // namespace

#include "sequence_engine.h"
#include "rig_dense.h"

extern "C" {

void rgbd360_default_params(rgbd360_params* p) {
    p->n_pyr = 4;
    p->min_depth = 0.3f;
    p->max_depth = 6.0f;
    p->sigma_photo = (float)(6. / 255);
    p->sigma_depth = (float)0.2;
    p->thres_sal_photo = 0.01f;
    p->thres_sal_depth = 0.01f;
    p->max_iters = 10;
    p->tol_residual = 1e-3f;
    p->tol_update = 1e-4f;
    p->mask_seams = 1;
    p->device = 0;
}

int rgbd360_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rgbd360_create(const rgbd360_params* p, rgbd360_ctx** out) {
    if (!p || !out) return -1;
    *out = nullptr;
    if (p->n_pyr < 1 || p->n_pyr > 8) return -1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return -100;   // no HIP device: no fallback
    if (p->device < 0 || p->device >= ndev) return -101;
    if (hipSetDevice(p->device) != hipSuccess) return -102;
    rgbd360_ctx* ctx = new rgbd360_ctx();
    ctx->p = *p;
    if (const char* e = knobs::debug("RGBD360_EVAL_BLOCKS")) {
        const int v = atoi(e);
        if (v >= 8 && v <= 8192) ctx->max_eval_blocks = v;
    }
    if (const char* e = knobs::debug("RGBD360_POLL_CHUNK")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 16) ctx->poll_chunk = v;
    }
    if (const char* e = knobs::debug("RGBD360_FIRST_CHUNK")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 16) ctx->first_chunk_top = v;
    }
    if (const char* e = knobs::debug("RGBD360_L0_CHUNK")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 16) ctx->chunk_level0 = v;
    }
    if (const char* e = knobs::debug("RGBD360_ADAPTIVE_CHUNKS")) {
        ctx->adaptive_chunks = atoi(e) != 0;
    }
    if (const char* e = knobs::debug("RGBD360_FUSED_SOLVE")) {
        ctx->fused_solve = atoi(e) != 0;
    }
    if (const char* e = knobs::debug("RGBD360_FUSED_OCC")) {
        ctx->fused_occ = atoi(e) != 0;
    }
    bool ok = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&ctx->ev0) == hipSuccess && hipEventCreate(&ctx->ev1) == hipSuccess &&
              hipMalloc(&ctx->d_state, sizeof(GNState)) == hipSuccess &&
              hipMemset(ctx->d_state, 0, sizeof(GNState)) == hipSuccess &&
              hipMalloc(&ctx->d_state_alt, sizeof(GNState)) == hipSuccess &&
              hipMemset(ctx->d_state_alt, 0, sizeof(GNState)) == hipSuccess &&
              hipMalloc(&ctx->d_gnio, sizeof(GnIO)) == hipSuccess &&
              hipHostMalloc((void**)&ctx->h_state, sizeof(GNState), hostwait::kPublishedFlags) == hipSuccess &&
              hostwait::spin_tag_init(&ctx->tag) == hipSuccess;
    if (!ok) {
        rgbd360_destroy(ctx);
        return -103;
    }
    *out = ctx;
    return 0;
}

void rgbd360_destroy(rgbd360_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->p.device);
    for (rgbd360_ctx* sib : ctx->siblings) rgbd360_destroy(sib);
    ctx->siblings.clear();
    for (SeqEngine* e : ctx->engines) seq_free(e);
    ctx->engines.clear();
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    free_levels(ctx);
    hipFree(ctx->d_state); hipFree(ctx->d_state_alt); hipFree(ctx->d_partials); hipFree(ctx->d_partials_alt); hipFree(ctx->d_gnio);
    for (int k = 0; k < 2; ++k) {
        hipFree(ctx->d_stage_rgb[k]); hipFree(ctx->d_stage_depth[k]);
        if (ctx->up_ev[k]) hipEventDestroy(ctx->up_ev[k]);
        if (ctx->conv_ev[k]) hipEventDestroy(ctx->conv_ev[k]);
    }
    if (ctx->up_stream) { hipStreamSynchronize(ctx->up_stream); hipStreamDestroy(ctx->up_stream); }
    if (ctx->f360) f360_state_destroy(ctx->f360);
    hipFree(ctx->occ_head); hipFree(ctx->occ_nodes); hipFree(ctx->occ_runinfo);
    if (ctx->h_state) hipHostFree(ctx->h_state);
    hostwait::spin_tag_free(&ctx->tag);
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* rgbd360_last_error(rgbd360_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
void* rgbd360_stream(rgbd360_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
int rgbd360_sync(rgbd360_ctx* ctx) {
    if (!ctx) return -1;
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int rgbd360_set_target(rgbd360_ctx* ctx, const uint8_t* rgb, size_t rgb_step, const void* depth, size_t depth_step,
                       int depth_type, int rows, int cols) {
    if (ctx) hipSetDevice(ctx->p.device);
    return set_frame(ctx, true, rgb, rgb_step, depth, depth_step, depth_type, rows, cols, false);
}
int rgbd360_set_source(rgbd360_ctx* ctx, const uint8_t* rgb, size_t rgb_step, const void* depth, size_t depth_step,
                       int depth_type, int rows, int cols) {
    if (ctx) hipSetDevice(ctx->p.device);
    return set_frame(ctx, false, rgb, rgb_step, depth, depth_step, depth_type, rows, cols, false);
}
int rgbd360_set_target_dev(rgbd360_ctx* ctx, const uint8_t* rgb, size_t rgb_step, const void* depth, size_t depth_step,
                           int depth_type, int rows, int cols) {
    if (ctx) hipSetDevice(ctx->p.device);
    return set_frame(ctx, true, rgb, rgb_step, depth, depth_step, depth_type, rows, cols, true);
}
int rgbd360_set_source_dev(rgbd360_ctx* ctx, const uint8_t* rgb, size_t rgb_step, const void* depth, size_t depth_step,
                           int depth_type, int rows, int cols) {
    if (ctx) hipSetDevice(ctx->p.device);
    return set_frame(ctx, false, rgb, rgb_step, depth, depth_step, depth_type, rows, cols, true);
}

int rgbd360_promote_source_to_target(rgbd360_ctx* ctx) {
    if (!ctx) return -1;
    if (!ctx->have_src) return fail(ctx, -2, "no source frame to promote");
    hipSetDevice(ctx->p.device);
    for (Level& L : ctx->levels) {
        std::swap(L.graySrc, L.grayTrg);
        std::swap(L.depthSrc, L.depthTrg);
    }
    launch_gradient_recs(ctx);
    HIPC(ctx, hipGetLastError());
    ctx->have_trg = true;
    ctx->have_src = false;
    return 0;
}

// The whole coarse-to-fine schedule is enqueued ahead of the device: every launch carries its level and turns into a
// no-op unless that level is the active, unfinished one (k_level_init of level l only fires once level l+1 has
// finished).  In the common case -- each level converges within its first chunk -- the host synchronises ONCE per
// alignment; a level that needs more passes gets another chunk, followed again by the finer levels.
// The last solve launch of the schedule publishes the state to the host (SolveCfg::host_state): no copy launch behind it.
static void enqueue_schedule(rgbd360_ctx* ctx, int pending, bool pending_started) {
    const int top = ctx->p.n_pyr - 1;
    for (int level = pending; level >= 0; --level) {
        const bool fused = fused_ok(ctx, ctx->al_occ);
        const bool start = level == top && !pending_started;      // the finer levels are entered by the solve itself when a level finishes
        if (start && !fused) launch_level_init(ctx, level, ctx->al_guess, 1);      // (the fused schedule's first launch initialises the state itself)
        int n_pairs = (level == top && !pending_started) ? ctx->first_chunk_top : (level == 0 ? ctx->chunk_level0 : ctx->poll_chunk);
        // Consecutive alignments of a sequence take much the same number of iterations per level: the first visit of a level is
        // given what the previous alignment needed there (its accepted iterations + the first pass + the pass that ends the level: one
        // launch fewer costs a host round trip per level, 173 -> 293 us measured)
        // instead of the fixed chunk -- fewer no-op launches on the coarse levels (~3 us each), no second round trip on level 0.
        if (ctx->adaptive_chunks && !(level == pending && pending_started) && ctx->hist_iters[level & 7] >= 0)
            n_pairs = std::min(std::max(ctx->hist_iters[level & 7] + 2, 2), 12);
        if (fused) {
            // one launch per iteration: n_pairs passes, each carrying the solve of the one before it; the solve of the chunk's last
            // pass rides in the next level's first launch, the schedule's very last one in a one-block launch that publishes
            for (int k = 0; k < n_pairs; ++k) {
                const float* init_pose = start && k == 0 ? ctx->al_guess : nullptr;
                if (ctx->al_occ) launch_occ_fused(ctx, level, ctx->al_method, ctx->al_occ, 0, init_pose);
                else launch_eval_fused(ctx, level, ctx->al_method, 0, init_pose);
            }
            if (level == 0) launch_solve_pending(ctx, 0, /*publish=*/true, ctx->al_occ);
            continue;
        }
        for (int k = 0; k < n_pairs; ++k) {
            launch_eval(ctx, level, ctx->al_method, true, ctx->al_occ);
            launch_solve(ctx, level, 0, 0, ctx->al_occ, /*publish=*/level == 0 && k == n_pairs - 1);
        }
    }
}

int rgbd360_align360_begin(rgbd360_ctx* ctx, const float guess[16], int method, int occlusion) {
    int rc = check_args(ctx, 0, method);
    if (rc) return rc;
    if (occlusion < 0 || occlusion > 2) return fail(ctx, -5, "occlusion must be 0, 1 or 2");
    if (!guess) return fail(ctx, -1, "null pose pointer");
    hipSetDevice(ctx->p.device);
    if (occlusion != 0 && (rc = occ_ensure(ctx)) != 0) return rc;
    ctx->al_occ = occlusion;
    memcpy(ctx->al_guess, guess, sizeof(ctx->al_guess));
    ctx->al_method = method;
    ctx->al_active = true;
    enqueue_schedule(ctx, ctx->p.n_pyr - 1, false);      // (its last solve publishes the state and bumps the tag)
    HIPC(ctx, hipGetLastError());
    return 0;
}

int rgbd360_align360_finish(rgbd360_ctx* ctx, float pose_out[16], rgbd360_result* res) {
    if (!ctx || !pose_out) return -1;
    if (!ctx->al_active) return fail(ctx, -2, "rgbd360_align360_begin was not called");
    hipSetDevice(ctx->p.device);
    ctx->al_active = false;
    for (int round = 0;; ++round) {
        HIPC(ctx, hostwait::wait(ctx->tag, ctx->stream));      // the state published by begin / the previous round has landed
        const GNState& S = *ctx->h_state;
        if (S.status != 0) break;
        if (S.level_active == 0 && S.done) break;
        if (S.done) return fail(ctx, -6, "alignment schedule stalled between levels");
        if (round > (ctx->p.max_iters + 4) * ctx->p.n_pyr) return fail(ctx, -6, "alignment loop did not terminate");
        enqueue_schedule(ctx, S.level_active, true);         // the stalled level gets another chunk, then the finer ones
        HIPC(ctx, hipGetLastError());
    }
    rgbd360_result R;
    result_from_state(*ctx->h_state, ctx->p.n_pyr, ctx->al_occ, pose_out, &R);
    for (int l = 0; l < 8; ++l) ctx->hist_iters[l] = (R.status == 0 && l < ctx->p.n_pyr) ? ctx->h_state->iters[l] : -1;
    if (res) *res = R;
    return R.status;
}

int rgbd360_align360(rgbd360_ctx* ctx, const float guess[16], int method, int occlusion, float pose_out[16],
                     rgbd360_result* res) {
    if (!pose_out) return ctx ? fail(ctx, -1, "null pose pointer") : -1;
    const int rc = rgbd360_align360_begin(ctx, guess, method, occlusion);
    if (rc) return rc;
    return rgbd360_align360_finish(ctx, pose_out, res);
}

// ---------------------------------------------------------------------------------------------------------
// A sequence of consecutive pairs on one GPU (SURVEY.md 8b rgbd360_align360_batch, 8e): pair j = (frame j, frame j+1),
// frame j = target, frame j+1 = source, as OdometryRGBD360.cpp:141-297 walks a sequence.  The pairs are cut into
// n_inflight contiguous sub-chunks, one context (own HIP stream) each; inside a sub-chunk frame j+1 is uploaded once and
// promoted from source to target; in every step all live contexts are enqueued before any is waited for.
// ---------------------------------------------------------------------------------------------------------
static int align360_batch_threads(rgbd360_ctx* ctx, int n_frames, const uint8_t* const* rgb, size_t rgb_step, const void* const* depth,
                               size_t depth_step, int depth_type, int rows, int cols, const float guess[16], int method,
                               int occlusion, int n_inflight, float* poses_out, rgbd360_result* results_out, bool on_device) {
    if (!ctx) return -1;
    if (!rgb || !depth || !poses_out) return fail(ctx, -1, "null pointer");
    if (n_frames < 1) return fail(ctx, -1, "n_frames must be >= 1");
    if (n_inflight < 1 || n_inflight > 16) return fail(ctx, -1, "n_inflight must be in 1..16");
    if (method < 0 || method > 2) return fail(ctx, -4, "bad method");
    if (occlusion < 0 || occlusion > 2) return fail(ctx, -5, "occlusion must be 0, 1 or 2");
    const int n = n_frames - 1;
    if (n == 0) return 0;
    for (int k = 0; k < n_frames; ++k)
        if (!rgb[k] || !depth[k]) return fail(ctx, -1, "null frame pointer");
    static const float kIdentity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    const float* g = guess ? guess : kIdentity;
    const int k_ctx = std::min(n_inflight, n);
    while ((int)ctx->siblings.size() < k_ctx - 1) {
        rgbd360_ctx* sib = nullptr;
        const int rc = rgbd360_create(&ctx->p, &sib);
        if (rc) return fail(ctx, rc, "cannot create a sibling context");
        ctx->siblings.push_back(sib);
    }
    std::vector<rgbd360_ctx*> cs(k_ctx);
    cs[0] = ctx;
    for (int c = 1; c < k_ctx; ++c) {
        cs[c] = ctx->siblings[c - 1];
        if (cs[c]->index_libm != ctx->index_libm) rgbd360_set_index_arithmetic(cs[c], ctx->index_libm);      // the siblings follow the context's warp arithmetic
    }
    // contiguous balanced spans [a, b) of pairs per context
    std::vector<int> a(k_ctx), b(k_ctx);
    for (int c = 0; c < k_ctx; ++c) {
        const int base = n / k_ctx, extra = n % k_ctx;
        a[c] = c * base + std::min(c, extra);
        b[c] = a[c] + base + (c < extra ? 1 : 0);
    }
    // One host thread per sub-chunk: a 4-level alignment is ~65 dependent launches, so a single enqueueing thread (≈2.7 us
    // per launch) caps a GPU at ≈5.7 k alignments/s however many contexts are in flight.  The sub-chunks share nothing
    // (own context, stream, buffers, error string; disjoint output slots), so each walks its pairs on its own thread:
    // upload (asynchronous, waited for only by the pair's own finish) -> frame set-up -> schedule -> finish -> promote.
    std::vector<int> rcs(k_ctx, 0);
    auto run_chunk = [&](int c) {
        rgbd360_ctx* cc = cs[c];
        hipSetDevice(cc->p.device);
        int rc = 0;
        if (on_device) {
            rc = set_frame(cc, true, rgb[a[c]], rgb_step, depth[a[c]], depth_step, depth_type, rows, cols, true);
        } else {        // frame f of this sub-chunk travels through staging slot (f - a) & 1, one frame ahead of the alignment
            rc = upload_stage(cc, 0, rgb[a[c]], rgb_step, depth[a[c]], depth_step, depth_type, rows, cols);
            if (!rc) rc = upload_stage(cc, 1, rgb[a[c] + 1], rgb_step, depth[a[c] + 1], depth_step, depth_type, rows, cols);
            if (!rc) rc = set_frame(cc, true, nullptr, 0, nullptr, 0, depth_type, rows, cols, false, false, 0);
        }
        for (int j = a[c]; !rc && j < b[c]; ++j) {
            if (on_device) {
                rc = set_frame(cc, false, rgb[j + 1], rgb_step, depth[j + 1], depth_step, depth_type, rows, cols, true);
            } else {
                rc = set_frame(cc, false, nullptr, 0, nullptr, 0, depth_type, rows, cols, false, false, (j + 1 - a[c]) & 1);
                if (!rc && j + 2 <= b[c])       // the next source frame is copied while this pair is being aligned
                    rc = upload_stage(cc, (j + 2 - a[c]) & 1, rgb[j + 2], rgb_step, depth[j + 2], depth_step, depth_type, rows, cols);
            }
            if (!rc) rc = rgbd360_align360_begin(cc, g, method, occlusion);
            if (rc) break;
            rgbd360_result R;
            rc = rgbd360_align360_finish(cc, poses_out + (size_t)16 * j, &R);      // >= 0: the pair's outcome, kept in R.status
            if (results_out) results_out[j] = R;
            if (rc >= 0) rc = j + 1 < b[c] ? std::min(0, rgbd360_promote_source_to_target(cc)) : 0;
        }
        if (cc->up_stream) hipStreamSynchronize(cc->up_stream);      // no upload may outlive the caller's buffers
        if (rc) hipStreamSynchronize(cc->stream);
        rcs[c] = rc;
    };
    {
        std::vector<std::thread> workers;
        workers.reserve(k_ctx - 1);
        std::vector<int> inline_chunks;              // sub-chunks whose thread could not be started run on the caller's thread
        for (int c = 1; c < k_ctx; ++c) {
            try {
                workers.emplace_back(run_chunk, c);
            } catch (const std::exception&) {        // no exception may cross the C boundary
                inline_chunks.push_back(c);
            }
        }
        run_chunk(0);
        for (int c : inline_chunks) run_chunk(c);
        for (std::thread& w : workers) w.join();
    }
    for (int c = 0; c < k_ctx; ++c)
        if (rcs[c]) return cs[c] == ctx ? rcs[c] : fail(ctx, rcs[c], cs[c]->err.c_str());
    return 0;
}

// The lock-step route (sequence_engine.h): n_inflight = pairs in flight = slots, spread over one or two engines (own stream and host
// thread each: while one engine waits for its round's states or enqueues, the other's kernels fill the device).
// Contexts of the per-context route.  The device runs four hardware queues side by side; a fifth busy queue makes the command
// processor time-slice them, and throughput collapses (64 pairs at 1024 x 512, host frames: 6.8 k alignments/s with 3 contexts,
// 2.2 k with 4, when GPU_MAX_HW_QUEUES=8 gives every stream a queue of its own).  With the runtime's default of 4 queues the streams
// share queues instead and 6 contexts are the optimum (7.6 k; 3: 6.5 k; 16: 6.6 k) -- tools/ctx_threads_perf.py.
// RGBD360_CTX_ROUTE_CAP overrides (measurements).
static int ctx_route_cap() {
    static const int v = [] {
        const char* q = knobs::product("GPU_MAX_HW_QUEUES");
        const int dflt = (q && atoi(q) > 4) ? 3 : 6;
        const char* e = knobs::debug("RGBD360_CTX_ROUTE_CAP");
        const int c = e ? atoi(e) : dflt;
        return c >= 1 && c <= 16 ? c : dflt;
    }();
    return v;
}

static int align360_batch_lockstep(rgbd360_ctx* ctx, int n_frames, const uint8_t* const* rgb, size_t rgb_step, const void* const* depth,
                                   size_t depth_step, int depth_type, int rows, int cols, const float* g, int method, int n_inflight,
                                   float* poses_out, rgbd360_result* results_out, bool on_device) {
    const int n = n_frames - 1;
    // host frames: the call is PCIe-bound and a round's frames travel while the previous round is aligned -- more than 16 slots only
    // lengthen the first (unoverlapped) upload and the staging buffers (4.37 k alignments/s with 16 slots, 4.12 k with 32)
    const int S = std::min(on_device ? n_inflight : std::min(n_inflight, 16), n);
    int n_eng = S >= 4 ? 2 : 1;
    if (const char* e = knobs::product("RGBD360_SEQ_ENGINES")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 4) n_eng = std::min(v, S);
    }
    while ((S + n_eng - 1) / n_eng > kMaxSlots) ++n_eng;
    std::vector<int> cnt(n_eng), off(n_eng);
    for (int e = 0, o = 0; e < n_eng; ++e) {
        cnt[e] = S / n_eng + (e < S % n_eng ? 1 : 0);
        off[e] = o;
        o += cnt[e];
    }
    hipSetDevice(ctx->p.device);
    // engines are kept between calls; geometry or slot count changes rebuild them
    bool rebuild = (int)ctx->engines.size() != n_eng;
    for (int e = 0; !rebuild && e < n_eng; ++e)
        rebuild = ctx->engines[e]->rows != rows || ctx->engines[e]->cols != cols || ctx->engines[e]->P != cnt[0];
    if (rebuild) {
        for (SeqEngine* e : ctx->engines) seq_free(e);
        ctx->engines.clear();
        for (int e = 0; e < n_eng; ++e) {
            SeqEngine* E = nullptr;
            std::string err;
            const int rc = seq_create(ctx->p, cnt[0], rows, cols, ctx->max_eval_blocks, &E, &err);
            if (rc == 0) E->libm = ctx->index_libm;
            if (rc) return fail(ctx, rc, err.c_str());
            ctx->engines.push_back(E);
        }
    }
    std::vector<int> a(S), b(S);      // contiguous balanced spans [a, b) of pairs per slot
    for (int c = 0; c < S; ++c) {
        const int base = n / S, extra = n % S;
        a[c] = c * base + std::min(c, extra);
        b[c] = a[c] + base + (c < extra ? 1 : 0);
    }
    std::vector<int> rcs(n_eng, 0);
    auto run_engine = [&](int e) {
        rcs[e] = seq_run(ctx->engines[e], cnt[e], a.data() + off[e], b.data() + off[e], rgb, rgb_step, depth, depth_step, depth_type, g, method,
                         on_device, poses_out, results_out);
        if (rcs[e]) hipStreamSynchronize(ctx->engines[e]->stream);
    };
    {
        std::vector<std::thread> workers;
        std::vector<int> inline_engines;
        for (int e = 1; e < n_eng; ++e) {
            try {
                workers.emplace_back(run_engine, e);
            } catch (const std::exception&) {        // no exception may cross the C boundary
                inline_engines.push_back(e);
            }
        }
        run_engine(0);
        for (int e : inline_engines) run_engine(e);
        for (std::thread& w : workers) w.join();
    }
    for (int e = 0; e < n_eng; ++e)
        if (rcs[e]) return fail(ctx, rcs[e], ctx->engines[e]->err.c_str());
    return 0;
}

static int align360_batch_impl(rgbd360_ctx* ctx, int n_frames, const uint8_t* const* rgb, size_t rgb_step, const void* const* depth,
                               size_t depth_step, int depth_type, int rows, int cols, const float guess[16], int method,
                               int occlusion, int n_inflight, float* poses_out, rgbd360_result* results_out, bool on_device) {
    if (!ctx) return -1;
    if (!rgb || !depth || !poses_out) return fail(ctx, -1, "null pointer");
    if (n_frames < 1) return fail(ctx, -1, "n_frames must be >= 1");
    if (n_inflight < 1 || n_inflight > 64) return fail(ctx, -1, "n_inflight must be in 1..64");
    if (method < 0 || method > 2) return fail(ctx, -4, "bad method");
    if (occlusion < 0 || occlusion > 2) return fail(ctx, -5, "occlusion must be 0, 1 or 2");
    if (depth_type != 0 && depth_type != 1) return fail(ctx, -1, "depth_type must be 0 (u16 mm) or 1 (f32 m)");
    if (n_frames == 1) return 0;
    for (int k = 0; k < n_frames; ++k)
        if (!rgb[k] || !depth[k]) return fail(ctx, -1, "null frame pointer");
    static const float kIdentity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    const char* route = knobs::debug("RGBD360_SEQ_ROUTE");          // "contexts": the per-context route for every sequence (A/B measurements)
    // the occlusion-aware passes have no slot dimension: those sequences run one context per sub-chunk
    if (occlusion != 0 || ctx->seq_route_contexts || (route && strcmp(route, "contexts") == 0))
        // (capped: more busy streams than hardware queues run side by side costs a factor of three, see ctx_route_cap)
        return align360_batch_threads(ctx, n_frames, rgb, rgb_step, depth, depth_step, depth_type, rows, cols, guess, method, occlusion,
                                      std::min(n_inflight, ctx->seq_route_cap > 0 ? ctx->seq_route_cap : ctx_route_cap()), poses_out, results_out, on_device);
    return align360_batch_lockstep(ctx, n_frames, rgb, rgb_step, depth, depth_step, depth_type, rows, cols, guess ? guess : kIdentity, method,
                                   n_inflight, poses_out, results_out, on_device);
}

int rgbd360_align360_batch(rgbd360_ctx* ctx, int n_frames, const uint8_t* const* rgb, size_t rgb_step, const void* const* depth,
                           size_t depth_step, int depth_type, int rows, int cols, const float guess[16], int method,
                           int occlusion, int n_inflight, float* poses_out, rgbd360_result* results_out) {
    return align360_batch_impl(ctx, n_frames, rgb, rgb_step, depth, depth_step, depth_type, rows, cols, guess, method, occlusion,
                               n_inflight, poses_out, results_out, false);
}
int rgbd360_align360_batch_dev(rgbd360_ctx* ctx, int n_frames, const uint8_t* const* rgb, size_t rgb_step, const void* const* depth,
                               size_t depth_step, int depth_type, int rows, int cols, const float guess[16], int method,
                               int occlusion, int n_inflight, float* poses_out, rgbd360_result* results_out) {
    return align360_batch_impl(ctx, n_frames, rgb, rgb_step, depth, depth_step, depth_type, rows, cols, guess, method, occlusion,
                               n_inflight, poses_out, results_out, true);
}

int rgbd360_level_dims(rgbd360_ctx* ctx, int level, int* rows, int* cols) {
    if (!ctx || level < 0 || level >= (int)ctx->levels.size()) return -3;
    *rows = ctx->levels[level].rows;
    *cols = ctx->levels[level].cols;
    return 0;
}

int rgbd360_get_plane(rgbd360_ctx* ctx, int which, int level, float* host_out) {
    if (!ctx || !host_out) return -1;
    if (level < 0 || level >= (int)ctx->levels.size()) return fail(ctx, -3, "bad pyramid level");
    hipSetDevice(ctx->p.device);
    const Level& L = ctx->levels[level];
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    if (which >= 0 && which <= 3) {
        const float* src = which == 0 ? L.graySrc : which == 1 ? L.grayTrg : which == 2 ? L.depthSrc : L.depthTrg;
        HIPC(ctx, hipMemcpy(host_out, src, (size_t)L.n * sizeof(float), hipMemcpyDeviceToHost));
        return 0;
    }
    if (which >= 4 && which <= 7) {
        std::vector<F3> tmp(L.n);
        HIPC(ctx, hipMemcpy(tmp.data(), which < 6 ? L.trgP : L.trgD, (size_t)L.n * sizeof(F3), hipMemcpyDeviceToHost));
        const bool x = (which == 4 || which == 6);
        for (int i = 0; i < L.n; ++i) host_out[i] = x ? tmp[i].b : tmp[i].c;
        return 0;
    }
    return fail(ctx, -4, "bad plane id");
}

int rgbd360_get_lut(rgbd360_ctx* ctx, int level, float* host_out_xyz) {
    if (!ctx || !host_out_xyz) return -1;
    if (level < 0 || level >= (int)ctx->levels.size()) return fail(ctx, -3, "bad pyramid level");
    if (!ctx->have_src) return fail(ctx, -2, "no source frame");
    hipSetDevice(ctx->p.device);
    const Level& L = ctx->levels[level];
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<float4> tmp(L.n);
    HIPC(ctx, hipMemcpy(tmp.data(), L.srcRec, (size_t)L.n * sizeof(float4), hipMemcpyDeviceToHost));
    for (int i = 0; i < L.n; ++i) {
        host_out_xyz[3 * i] = tmp[i].x;
        host_out_xyz[3 * i + 1] = tmp[i].y;
        host_out_xyz[3 * i + 2] = tmp[i].z;
    }
    return 0;
}

int rgbd360_eval(rgbd360_ctx* ctx, int level, const float pose[16], int method, double* err2, long long* n_valid,
                 double err2_split[2], long long n_split[2], float H[36], float g[6], double H64[36], double g64[6],
                 long long* n_visible) {
    return rgbd360_eval_occ(ctx, level, pose, method, 0, err2, n_valid, err2_split, n_split, H, g, H64, g64, n_visible);
}

int rgbd360_eval_occ(rgbd360_ctx* ctx, int level, const float pose[16], int method, int occlusion, double* err2,
                     long long* n_valid, double err2_split[2], long long n_split[2], float H[36], float g[6], double H64[36],
                     double g64[6], long long* n_visible) {
    int rc = check_args(ctx, level, method);
    if (rc) return rc;
    if (!pose) return fail(ctx, -1, "null pose pointer");
    if (occlusion < 0 || occlusion > 2) return fail(ctx, -5, "occlusion must be 0, 1 or 2");
    hipSetDevice(ctx->p.device);
    if (occlusion != 0 && (rc = occ_ensure(ctx)) != 0) return rc;
    launch_level_init(ctx, level, pose, 1);
    launch_eval(ctx, level, method, true, occlusion);
    launch_solve(ctx, level, 1, 0, occlusion);
    HIPC(ctx, hipGetLastError());
    rc = read_state(ctx);
    if (rc) return rc;
    const GNState& S = *ctx->h_state;
    if (err2) *err2 = S.tot[P_E2P] + S.tot[P_E2D];
    if (n_valid) *n_valid = (long long)(S.tot[P_NP] + S.tot[P_ND]);
    if (err2_split) { err2_split[0] = S.tot[P_E2P]; err2_split[1] = S.tot[P_E2D]; }
    if (n_split) { n_split[0] = (long long)S.tot[P_NP]; n_split[1] = (long long)S.tot[P_ND]; }
    if (H) memcpy(H, S.H, sizeof(float) * 36);
    if (g) memcpy(g, S.g, sizeof(float) * 6);
    if (H64 || g64) {
        int k = 0;
        for (int a = 0; a < 6; ++a)
            for (int b = a; b < 6; ++b, ++k)
                if (H64) H64[b * 6 + a] = H64[a * 6 + b] = S.tot[P_H + k];
        if (g64)
            for (int a = 0; a < 6; ++a) g64[a] = S.tot[P_G + a];
    }
    if (n_visible) *n_visible = (long long)S.tot[P_NVIS];
    return 0;
}

int rgbd360_warp_indices(rgbd360_ctx* ctx, int level, const float pose[16], int32_t* host_out_rc) {
    if (!ctx || !pose || !host_out_rc) return -1;
    if (level < 0 || level >= (int)ctx->levels.size()) return fail(ctx, -3, "bad pyramid level");
    if (!ctx->have_src) return fail(ctx, -2, "no source frame");
    hipSetDevice(ctx->p.device);
    const Level& L = ctx->levels[level];
    int32_t* d_out = nullptr;
    HIPC(ctx, hipMalloc(&d_out, (size_t)L.n * 2 * sizeof(int32_t)));
    Pose16 P;
    memcpy(P.v, pose, sizeof(P.v));
    hipLaunchKernelGGL(k_warp_indices, dim3((L.n + 255) / 256), dim3(256), 0, ctx->stream, level_dev(L), P, d_out);
    hipError_t e = hipMemcpyAsync(host_out_rc, d_out, (size_t)L.n * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(d_out);
    HIPC(ctx, e);
    return 0;
}

int rgbd360_gn_step(rgbd360_ctx* ctx, const float H[36], const float g[6], float lambda, const float pose[16],
                    float pose_tmp[16], float update[6]) {
    if (!ctx || !H || !g || !pose) return -1;
    hipSetDevice(ctx->p.device);
    GnIO io;
    memcpy(io.H, H, sizeof(io.H));
    memcpy(io.g, g, sizeof(io.g));
    memcpy(io.pose, pose, sizeof(io.pose));
    io.lambda = lambda;
    io.status = -1;
    HIPC(ctx, hipMemcpyAsync(ctx->d_gnio, &io, sizeof(io), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_gn_step, dim3(1), dim3(64), 0, ctx->stream, ctx->d_gnio);
    HIPC(ctx, hipMemcpyAsync(&io, ctx->d_gnio, sizeof(io), hipMemcpyDeviceToHost, ctx->stream));
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    if (pose_tmp) memcpy(pose_tmp, io.pose_tmp, sizeof(io.pose_tmp));
    if (update) memcpy(update, io.update, sizeof(io.update));
    return io.status;
}

int rgbd360_forced_iters(rgbd360_ctx* ctx, int level, const float pose0[16], int method, int n_iters, float pose_out[16],
                         double* last_rms, float* elapsed_ms) {
    int rc = check_args(ctx, level, method);
    if (rc) return rc;
    if (!pose0 || n_iters < 1) return fail(ctx, -1, "bad arguments");
    hipSetDevice(ctx->p.device);
    const bool fold_init = fused_ok(ctx, 0) && !elapsed_ms;      // the first fused launch initialises the state itself (timed calls keep
    if (!fold_init) launch_level_init(ctx, level, pose0, 1);      // the initialisation outside the events)
    if (elapsed_ms) HIPC(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    if (fused_ok(ctx, 0)) {       // n_iters launches {solve of the previous pass, pass} + the last solve
        for (int k = 0; k < n_iters; ++k) launch_eval_fused(ctx, level, method, 1, fold_init && k == 0 ? pose0 : nullptr);
        launch_solve_pending(ctx, 1, /*publish=*/!elapsed_ms);
    } else {
        for (int k = 0; k < n_iters; ++k) {
            launch_eval(ctx, level, method, true);
            launch_solve(ctx, level, 0, 1, 0, /*publish=*/!elapsed_ms && k == n_iters - 1);
        }
    }
    if (elapsed_ms) HIPC(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIPC(ctx, hipGetLastError());
    // events want the synchronise; without them the last solve has published the state and the host spins on its tag (host_wait.h)
    if (elapsed_ms) rc = read_state_sync(ctx);
    else HIPC(ctx, hostwait::wait(ctx->tag, ctx->stream));
    if (rc) return rc;
    if (elapsed_ms) HIPC(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    if (pose_out) memcpy(pose_out, ctx->h_state->pose, sizeof(float) * 16);
    if (last_rms) *last_rms = ctx->h_state->error;
    return ctx->h_state->status;
}

// One solve on a hand-made partial table (row 0 = `row`, every other row zero) at the identity pose, through the two-launch form
// (k_solve) or through the fused form (the prologue of k_eval_fs; the state is read as that launch leaves it): the state-machine paths real images hardly
// ever reach (ILL-POSED by the rank test alone, by a zero pivot, ...) can be driven from a test.  Needs both frames set (the fused
// form runs a pass afterwards when the step is accepted).  out_i: {status, done, level_active, it, n_evals, pend_nb}.
int rgbd360_debug_solve_partials(rgbd360_ctx* ctx, int level, const double row[32], int method, int fused, int out_i[6], float cand_out[16],
                                 float update_out[6]) {
    int rc = check_args(ctx, level, method);
    if (rc) return rc;
    if (!row || !out_i) return fail(ctx, -1, "bad arguments");
    if (fused && !fused_ok(ctx, 0)) return fail(ctx, -1, "the fused-solve schedule is switched off");
    hipSetDevice(ctx->p.device);
    const Level& L = ctx->levels[level];
    float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    launch_level_init(ctx, level, I, 1);
    const size_t rows = (size_t)std::max((ctx->max_blocks + 31) / 32 * 32, kPendingRows);
    HIPC(ctx, hipMemsetAsync(ctx->d_partials, 0, rows * kNumPartials * sizeof(double), ctx->stream));
    // fused == 2: the row sits beyond the first batch of rows and the launch is given too small a bound -- stage_pending has to notice
    // that the state holds more rows than the host said and fetch them (the bound is checked on the device, not trusted)
    const int row_at = fused == 2 ? 40 : 0;
    if (row_at >= L.nblocks) return fail(ctx, -1, "the level has too few block rows for the late-row form");
    HIPC(ctx, hipMemcpyAsync(ctx->d_partials + (size_t)row_at * kNumPartials, row, kNumPartials * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (fused) {
        const int pend[2] = {L.nblocks, L.n};       // as if a pass of this level had just written the table
        HIPC(ctx, hipMemcpyAsync(&ctx->d_state->pend_nb, pend, sizeof(pend), hipMemcpyHostToDevice, ctx->stream));
        ctx->pend_rows_hint = fused == 2 ? 1 : kPendingRows;      // a hand-made pending pass: no bound known
        launch_eval_fused(ctx, level, method, 0);      // the state is read as this launch leaves it: its own pass (if it ran one) stays pending
    } else {
        launch_solve(ctx, level, 0, 0);
    }
    HIPC(ctx, hipGetLastError());
    rc = read_state_sync(ctx);
    if (rc) return rc;
    const GNState& S = *ctx->h_state;
    out_i[0] = S.status; out_i[1] = S.done; out_i[2] = S.level_active; out_i[3] = S.it; out_i[4] = S.n_evals; out_i[5] = S.pend_nb;
    if (cand_out) memcpy(cand_out, S.cand, sizeof(float) * 16);
    if (update_out) memcpy(update_out, S.update, sizeof(float) * 6);
    return 0;
}

int rgbd360_time_eval_kernel(rgbd360_ctx* ctx, int level, const float pose[16], int method, int want_hg, int reps,
                             float* avg_us) {
    int rc = check_args(ctx, level, method);
    if (rc) return rc;
    if (!pose || reps < 1 || !avg_us) return fail(ctx, -1, "bad arguments");
    hipSetDevice(ctx->p.device);
    launch_level_init(ctx, level, pose, 1);
    const bool fused = want_hg == 2;      // the product's single-pair launch: solve of the previous pass + pass (forced schedule)
    if (fused && !fused_ok(ctx, 0)) return fail(ctx, -1, "the fused-solve schedule is switched off");
    auto one = [&]() {
        if (fused) launch_eval_fused(ctx, level, method, 1);
        else launch_eval(ctx, level, method, want_hg != 0);
    };
    one();   // warm-up
    HIPC(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int k = 0; k < reps; ++k) one();
    HIPC(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    if (fused) launch_solve_pending(ctx, 1, false);
    HIPC(ctx, hipGetLastError());
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIPC(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *avg_us = ms * 1000.f / reps;
    return 0;
}

// The same timer with the launches ROTATING over n_ctx contexts (all on ctxs[0]'s device, each holding its own copy of a frame
// pair) on ctxs[0]'s stream: with n_ctx x working set > 256 MiB every launch finds its records evicted from the Infinity Cache,
// so the average is an HBM-fed launch -- the back-to-back timer above re-reads an Infinity-Cache-resident set at 2048x1024.
int rgbd360_time_eval_kernel_rotating(rgbd360_ctx* const* ctxs, int n_ctx, int level, const float pose[16], int method, int want_hg,
                                      int reps, float* avg_us) {
    if (!ctxs || n_ctx < 1 || !ctxs[0]) return -1;
    rgbd360_ctx* c0 = ctxs[0];
    if (!pose || reps < 1 || !avg_us) return fail(c0, -1, "bad arguments");
    for (int k = 0; k < n_ctx; ++k) {
        if (!ctxs[k]) return fail(c0, -1, "null context");
        const int rc = check_args(ctxs[k], level, method);
        if (rc) return k == 0 ? rc : fail(c0, rc, ctxs[k]->err.c_str());
        if (ctxs[k]->p.device != c0->p.device) return fail(c0, -1, "all contexts must live on one device");
        if (want_hg == 2 && !fused_ok(ctxs[k], 0)) return fail(c0, -1, "the fused-solve schedule is switched off");
    }
    hipSetDevice(c0->p.device);
    // every context's kernels go to c0's stream for the duration of the measurement: all contexts are synchronised first (the
    // only step that can fail), then the streams are swapped in a loop that cannot, and swapped back on every way out
    for (int k = 0; k < n_ctx; ++k) HIPC(c0, hipStreamSynchronize(ctxs[k]->stream));
    std::vector<hipStream_t> own(n_ctx);
    for (int k = 0; k < n_ctx; ++k) {
        own[k] = ctxs[k]->stream;
        ctxs[k]->stream = c0->stream;
    }
    const bool fused = want_hg == 2;      // k_eval_fs, forced schedule (every context iterates on its own pair)
    auto one = [&](rgbd360_ctx* c) {
        if (fused) launch_eval_fused(c, level, method, 1);
        else launch_eval(c, level, method, want_hg != 0);
    };
    for (int k = 0; k < n_ctx; ++k) {
        launch_level_init(ctxs[k], level, pose, 1);
        one(ctxs[k]);      // warm-up
    }
    hipError_t e = hipEventRecord(c0->ev0, c0->stream);
    for (int r = 0; r < reps && e == hipSuccess; ++r) one(ctxs[r % n_ctx]);
    if (e == hipSuccess) e = hipEventRecord(c0->ev1, c0->stream);
    if (fused) for (int k = 0; k < n_ctx; ++k) launch_solve_pending(ctxs[k], 1, false);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c0->stream);
    for (int k = 0; k < n_ctx; ++k) ctxs[k]->stream = own[k];
    HIPC(c0, e);
    float ms = 0.f;
    HIPC(c0, hipEventElapsedTime(&ms, c0->ev0, c0->ev1));
    *avg_us = ms * 1000.f / reps;
    return 0;
}

// The forced schedule of rgbd360_forced_iters in the lock-step engine's regime: n_pairs copies of ONE pair iterate side by side, every
// {pass, solve} launch serving all of them (k_eval_b / k_solve_b), n_iters Gauss-Newton iterations each on `level`.  What a GN
// iteration costs when the solve launch and the launch boundaries are shared by the pairs in flight -- the product's sequence path.
int rgbd360_forced_iters_batch(rgbd360_ctx* ctx, int n_pairs, const uint8_t* rgb_trg, const void* depth_trg, const uint8_t* rgb_src,
                               const void* depth_src, size_t rgb_step, size_t depth_step, int depth_type, int rows, int cols, int level,
                               const float pose0[16], int method, int n_iters, float* poses_out, float* elapsed_ms, float* pass_avg_us) {
    if (!ctx) return -1;
    if (!rgb_trg || !depth_trg || !rgb_src || !depth_src || !pose0 || n_iters < 1 || n_pairs < 1 || n_pairs > kMaxSlots)
        return fail(ctx, -1, "bad arguments");
    if (method < 0 || method > 2) return fail(ctx, -4, "bad method");
    if (level < 0 || level >= ctx->p.n_pyr) return fail(ctx, -3, "bad pyramid level");
    if (depth_type != 0 && depth_type != 1) return fail(ctx, -1, "depth_type must be 0 (u16 mm) or 1 (f32 m)");
    hipSetDevice(ctx->p.device);
    SeqEngine* E = nullptr;
    std::string err;
    int rc = seq_create(ctx->p, n_pairs, rows, cols, ctx->max_eval_blocks, &E, &err);
    if (rc == 0) E->libm = ctx->index_libm;
    if (rc) return fail(ctx, rc, err.c_str());
    const size_t dpx = depth_type == 0 ? 2 : 4;
    const size_t fr = (size_t)rows * cols * 3, fd = (size_t)rows * cols * dpx;
    uint8_t *d_rgb[2] = {nullptr, nullptr}, *d_dep[2] = {nullptr, nullptr};
    auto cleanup = [&]() {
        for (int k = 0; k < 2; ++k) { hipFree(d_rgb[k]); hipFree(d_dep[k]); }
        seq_free(E);
    };
    hipError_t e = hipSuccess;
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
        e = hipMalloc(&d_rgb[k], fr);
        if (e == hipSuccess) e = hipMalloc(&d_dep[k], fd);
        if (e == hipSuccess) e = hipMemcpy2D(d_rgb[k], (size_t)cols * 3, k == 0 ? rgb_trg : rgb_src, rgb_step, (size_t)cols * 3, rows, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy2D(d_dep[k], (size_t)cols * dpx, k == 0 ? depth_trg : depth_src, depth_step, (size_t)cols * dpx, rows, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) { cleanup(); return fail(ctx, -(int)e - 1000, hipGetErrorString(e)); }
    const unsigned long long live = n_pairs >= 64 ? ~0ull : ((1ull << n_pairs) - 1ull);
    FramePtrs fp;
    memset(&fp, 0, sizeof(fp));
    for (int s = 0; s < n_pairs; ++s) { fp.rgb[s] = d_rgb[0]; fp.depth[s] = d_dep[0]; }
    seq_frame_setup(E, fp, (size_t)cols * 3, (size_t)cols * dpx, depth_type, live, 0ull, live, E->tb);            // targets
    for (int s = 0; s < n_pairs; ++s) { fp.rgb[s] = d_rgb[1]; fp.depth[s] = d_dep[1]; }
    seq_frame_setup(E, fp, (size_t)cols * 3, (size_t)cols * dpx, depth_type, live, live, 0ull, E->tb ^ 1);         // sources
    Pose16 Pz;
    memcpy(Pz.v, pose0, sizeof(Pz.v));
    hipLaunchKernelGGL(k_level_init_b, dim3(n_pairs), dim3(64), 0, E->stream, E->d_states, Pz, 1, 1, level, live);
    for (int k = 0; k < 3; ++k) {        // warm
        seq_launch_eval(E, level, method);
        seq_launch_solve(E, level, 1);
    }
    hipLaunchKernelGGL(k_level_init_b, dim3(n_pairs), dim3(64), 0, E->stream, E->d_states, Pz, 1, 1, level, live);
    e = hipEventRecord(ctx->ev0, E->stream);
    for (int k = 0; k < n_iters; ++k) {
        seq_launch_eval(E, level, method);
        seq_launch_solve(E, level, 1);
    }
    if (e == hipSuccess) e = hipEventRecord(ctx->ev1, E->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(E->h_states, E->d_states, (size_t)n_pairs * sizeof(GNState), hipMemcpyDeviceToHost, E->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(E->stream);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    if (e == hipSuccess && pass_avg_us) {      // the batch pass alone: back-to-back launches at the poses the iterations reached
        constexpr int kReps = 10;
        e = hipEventRecord(ctx->ev0, E->stream);
        for (int k = 0; k < kReps; ++k) seq_launch_eval(E, level, method);
        if (e == hipSuccess) e = hipEventRecord(ctx->ev1, E->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(E->stream);
        float pms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&pms, ctx->ev0, ctx->ev1);
        if (e == hipSuccess) *pass_avg_us = pms * 1e3f / kReps;
    }
    int status = 0;
    if (e == hipSuccess) {
        if (elapsed_ms) *elapsed_ms = ms;
        for (int s = 0; s < n_pairs; ++s) {
            if (poses_out) memcpy(poses_out + 16 * s, E->h_states[s].pose, sizeof(float) * 16);
            if (E->h_states[s].status != 0) status = E->h_states[s].status;
        }
    }
    cleanup();
    if (e != hipSuccess) return fail(ctx, -(int)e - 1000, hipGetErrorString(e));
    return status;
}

int rgbd360_time_solve_kernel(rgbd360_ctx* ctx, int level, int mode, int reps, float* avg_us) {
    int rc = check_args(ctx, level, 0);
    if (rc) return rc;
    if (reps < 1 || !avg_us) return fail(ctx, -1, "bad arguments");
    hipSetDevice(ctx->p.device);
    launch_solve(ctx, level, mode, 1);   // warm-up (partials of the last pass)
    HIPC(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int k = 0; k < reps; ++k) launch_solve(ctx, level, mode, 1);
    HIPC(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIPC(ctx, hipGetLastError());
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIPC(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *avg_us = ms * 1000.f / reps;
    return 0;
}

int rgbd360_debug_eval_stamps(rgbd360_ctx* ctx, int level, double out[12]) {
    if (!ctx || !out || level < 0 || level >= (int)ctx->levels.size()) return -1;
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    const int nb = ctx->levels[level].nblocks;
    HIPC(ctx, hipMemcpy(out, ctx->d_partials + (size_t)nb * kNumPartials, 6 * sizeof(double), hipMemcpyDeviceToHost));
    HIPC(ctx, hipMemcpy(out + 6, ctx->d_partials + (size_t)(nb + 1) * kNumPartials, 6 * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int rgbd360_debug_eval_blocks(rgbd360_ctx* ctx, int level, double* out /*2*nblocks*/) {
    if (!ctx || !out || level < 0 || level >= (int)ctx->levels.size()) return -1;
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    const int nb = ctx->levels[level].nblocks;
    HIPC(ctx, hipMemcpy(out, ctx->d_partials + (size_t)(nb + 8) * kNumPartials, 2 * nb * sizeof(double), hipMemcpyDeviceToHost));
    return nb;
}

int rgbd360_debug_eval_waves(rgbd360_ctx* ctx, int level, double* out /*16*nblocks*/) {      // per wave: "loop done", 100 MHz ticks from its block's start
    if (!ctx || !out || level < 0 || level >= (int)ctx->levels.size()) return -1;
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    const int nb = ctx->levels[level].nblocks;
    HIPC(ctx, hipMemcpy(out, ctx->d_partials + (size_t)(nb + 32) * kNumPartials, 16 * nb * sizeof(double), hipMemcpyDeviceToHost));
    return nb;
}

int rgbd360_debug_eval_history(rgbd360_ctx* ctx, int level, int reset, double out[120]) {
    if (!ctx || level < 0 || level >= (int)ctx->levels.size()) return -1;
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    const int nb = ctx->levels[level].nblocks;
    if (reset) HIPC(ctx, hipMemset(ctx->d_partials + (size_t)(nb + 2) * kNumPartials, 0, 5 * kNumPartials * sizeof(double)));
    if (out) HIPC(ctx, hipMemcpy(out, ctx->d_partials + (size_t)(nb + 3) * kNumPartials, 120 * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int rgbd360_debug_solve_stamps(rgbd360_ctx* ctx, unsigned long long out[8]) {
    if (!ctx || !out) return -1;
    int rc = read_state(ctx);
    if (rc) return rc;
    memcpy(out, ctx->h_state->stamps, sizeof(unsigned long long) * 8);
    return 0;
}

int rgbd360_selftest_math(rgbd360_ctx* ctx, uint32_t first_bits, uint32_t count, unsigned long long mismatches[3]) {
    if (!ctx || !mismatches) return -1;
    hipSetDevice(ctx->p.device);
    unsigned long long* d = nullptr;
    HIPC(ctx, hipMalloc(&d, 3 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, 3 * sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_selftest_math, dim3(2048), dim3(256), 0, ctx->stream, first_bits, count, d);
        e = hipMemcpyAsync(mismatches, d, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(d);
    HIPC(ctx, e);
    return 0;
}

// libm_f32.h as the DEVICE compiles it against the C library this process links (the reference's asinf / atanf / roundf / atan2f): the floats
// first_bits .. first_bits + count - 1 through the three one-argument functions, `count` drawn pairs through atan2f.  mismatches[4].
int rgbd360_selftest_libm(rgbd360_ctx* ctx, uint32_t first_bits, uint32_t count, unsigned long long mismatches[4]) {
    if (!ctx || !mismatches) return -1;
    hipSetDevice(ctx->p.device);
    for (int k = 0; k < 4; ++k) mismatches[k] = 0;
    constexpr uint32_t kChunk = 1u << 22;
    float* d = nullptr;
    HIPC(ctx, hipMalloc(&d, 4 * (size_t)kChunk * sizeof(float)));
    std::vector<float> h(4 * (size_t)kChunk);
    auto differ = [](float a, float b) {
        uint32_t x, y;
        memcpy(&x, &a, 4); memcpy(&y, &b, 4);
        return x != y && !((x & 0x7fffffffu) > 0x7f800000u && (y & 0x7fffffffu) > 0x7f800000u);      // (any NaN equals any NaN)
    };
    hipError_t e = hipSuccess;
    int shown = 0;
    for (uint64_t done = 0; done < count && e == hipSuccess; done += kChunk) {
        const uint32_t n = (uint32_t)std::min<uint64_t>(kChunk, count - done), first = first_bits + (uint32_t)done;
        hipLaunchKernelGGL(k_selftest_libm, dim3(2048), dim3(256), 0, ctx->stream, first, n, d);
        e = hipMemcpyAsync(h.data(), d, 4 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) break;
        for (uint32_t k = 0; k < n; ++k) {
            float v;
            const uint32_t u = first + k;
            memcpy(&v, &u, 4);
            if ((u & 0x7fffffffu) <= 0x3fc00000u) {
                const bool bad = differ(h[k], asinf(v));
                mismatches[0] += bad;
                if (bad && shown < 6 && knobs::debug("RGBD360_SELFTEST_VERBOSE")) {      // (debug builds: the first mismatches on stderr)
                    ++shown;
                    fprintf(stderr, "[selftest_libm] asinf(%.9g = 0x%08x): device %.9g, library %.9g\n", (double)v, u, (double)h[k], (double)asinf(v));
                }
            }
            mismatches[1] += differ(h[(size_t)n + k], atanf(v));
            mismatches[2] += differ(h[2 * (size_t)n + k], roundf(v));
            float y, x;
            selftest_libm_pair(u, y, x);
            mismatches[3] += differ(h[3 * (size_t)n + k], atan2f(y, x));
        }
    }
    hipFree(d);
    HIPC(ctx, e);
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// Pinhole single-sensor alignment: RegisterPhotoICP::alignFrames, RPI.h:4254-4512.
// Per-pixel passes on the device (pinhole_kernels.h), Levenberg-Marquardt driver on the host.
// ---------------------------------------------------------------------------------------------------------
namespace {
PinK pin_level_K(const rgbd360_ctx* ctx, int level) {         // RPI.h:571-575
    const float scaleFactor = 1.0 / pow(2, level);
    return {ctx->cam[0] * scaleFactor, ctx->cam[1] * scaleFactor, ctx->cam[2] * scaleFactor, ctx->cam[3] * scaleFactor};
}

int pin_check(rgbd360_ctx* ctx, int level, int method) {
    int rc = check_args(ctx, level, method);
    if (rc) return rc;
    if (!ctx->have_cam) return fail(ctx, -2, "rgbd360_set_camera was not called");
    if (ctx->p.mask_seams) return fail(ctx, -5, "pinhole alignment needs params.mask_seams = 0 (the seam mask belongs to the spherical panorama)");
    return 0;
}

int pin_prepare_level(rgbd360_ctx* ctx, int level) {
    Level& L = ctx->levels[level];
    if (!L.srcRecPin) HIPC(ctx, hipMalloc(&L.srcRecPin, (size_t)L.n * sizeof(float4)));
    const PinK K = pin_level_K(ctx, level);
    const float inv_fx = 1. / K.fx, inv_fy = 1. / K.fy;
    hipLaunchKernelGGL(k_src_rec_pinhole, grid2d(L.rows, L.cols), dim3(256), 0, ctx->stream, L.depthSrc, L.graySrc, L.rows, L.cols, K,
                       inv_fx, inv_fy, ctx->p.min_depth, ctx->p.max_depth, L.srcRecPin);
    HIPC(ctx, hipGetLastError());
    return 0;
}

LevelDev pin_level_dev(const Level& L) {
    LevelDev d = level_dev(L);
    d.src = L.srcRecPin;
    return d;
}

// one fused pass at `pose`; the reduced sums land in ctx->h_state->tot
int pin_eval(rgbd360_ctx* ctx, int level, const float* pose, int method) {
    const Level& L = ctx->levels[level];
    const LevelDev lv = pin_level_dev(L);
    const PinK K = pin_level_K(ctx, level);
    const EvalConsts ec = eval_consts(ctx->p);
    // two launches per evaluation: the pass (pose by kernel argument) and the reduce-only solve, which publishes the sums to the
    // host itself (was: state initialisation + pass + solve + copy, 21 us per round trip)
    Pose16 P;
    memcpy(P.v, pose, sizeof(P.v));
    const dim3 g(L.nblocks), b(kEvalThreads);
    if (method == 0) hipLaunchKernelGGL((k_eval_pinhole<0>), g, b, 0, ctx->stream, lv, K, ec, P, ctx->d_partials, L.chunk, level, ctx->sal_thr);
    else if (method == 1) hipLaunchKernelGGL((k_eval_pinhole<1>), g, b, 0, ctx->stream, lv, K, ec, P, ctx->d_partials, L.chunk, level, ctx->sal_thr);
    else hipLaunchKernelGGL((k_eval_pinhole<2>), g, b, 0, ctx->stream, lv, K, ec, P, ctx->d_partials, L.chunk, level, ctx->sal_thr);
    launch_solve(ctx, level, 1, 0, 0, /*publish=*/true);
    HIPC(ctx, hipGetLastError());
    HIPC(ctx, hostwait::wait(ctx->tag, ctx->stream));
    return 0;
}

// The occlusion-aware evaluation (pinhole_kernels.h, second half): per-target arrival lists, one walk per target pixel in source-index
// order; the reduce-only solve publishes the sums like pin_eval's.
int pin_occ_ensure(rgbd360_ctx* ctx, size_t n) {
    if (ctx->pin_occ_n >= n) return 0;
    pin_occ_free(ctx);
    PinOccLists& Ls = ctx->pin_lists;
    HIPC(ctx, hipMalloc(&ctx->pin_keys, n * sizeof(unsigned)));
    HIPC(ctx, hipMalloc(&ctx->pin_vals, n * sizeof(unsigned)));
    HIPC(ctx, hipMalloc(&Ls.cnt, n * sizeof(int)));
    HIPC(ctx, hipMalloc(&Ls.box, n * sizeof(int4)));
    HIPC(ctx, hipMalloc(&Ls.slots, n * kPinShort * sizeof(unsigned)));
    // armed once; every walk re-arms the words of the target pixels it visited
    hipLaunchKernelGGL(k_pin_occ_arm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, Ls.cnt, Ls.box, (int)n);
    HIPC(ctx, hipGetLastError());
    HIPC(ctx, hipMalloc(&ctx->pin_partials, ((n + kPinWalkThreads - 1) / kPinWalkThreads) * kNumPartials * sizeof(double)));
    ctx->pin_occ_n = n;
    return 0;
}

int pin_eval_occ(rgbd360_ctx* ctx, int level, const float* pose, int method, int occ) {
    const Level& L = ctx->levels[level];
    int rc = pin_occ_ensure(ctx, (size_t)ctx->levels[0].n);
    if (rc) return rc;
    const LevelDev lv = pin_level_dev(L);
    const PinK K = pin_level_K(ctx, level);
    const EvalConsts ec = eval_consts(ctx->p);
    Pose16 P;
    memcpy(P.v, pose, sizeof(P.v));
    const dim3 gk((L.n + 255) / 256), bk(256);
    if (occ == 1) hipLaunchKernelGGL((k_pin_occ_keys<1>), gk, bk, 0, ctx->stream, lv, K, P, ctx->pin_keys, ctx->pin_vals, ctx->pin_lists);
    else hipLaunchKernelGGL((k_pin_occ_keys<2>), gk, bk, 0, ctx->stream, lv, K, P, ctx->pin_keys, ctx->pin_vals, ctx->pin_lists);
    const int nblk = (L.n + kPinWalkThreads - 1) / kPinWalkThreads;
    const dim3 gw(nblk), bw(kPinWalkThreads);
    if (method == 0) hipLaunchKernelGGL((k_pin_occ_walk<0>), gw, bw, 0, ctx->stream, lv, K, ec, P, ctx->pin_keys, ctx->pin_vals, ctx->pin_lists, ctx->pin_partials);
    else if (method == 1) hipLaunchKernelGGL((k_pin_occ_walk<1>), gw, bw, 0, ctx->stream, lv, K, ec, P, ctx->pin_keys, ctx->pin_vals, ctx->pin_lists, ctx->pin_partials);
    else hipLaunchKernelGGL((k_pin_occ_walk<2>), gw, bw, 0, ctx->stream, lv, K, ec, P, ctx->pin_keys, ctx->pin_vals, ctx->pin_lists, ctx->pin_partials);
    SolveCfg cfg;
    cfg.level = level; cfg.mode = 1; cfg.forced = 0; cfg.max_iters = ctx->p.max_iters; cfg.n_pixels = L.n;
    cfg.occ = 0;
    cfg.tol_residual = ctx->p.tol_residual; cfg.tol_update = ctx->p.tol_update;
    cfg.host_state = ctx->h_state;
    cfg.host_tag = ctx->tag.h;
    cfg.host_seq = ++ctx->tag.seq;
    hipLaunchKernelGGL(k_solve, dim3(1), dim3(kSolveThreads), 0, ctx->stream, ctx->d_state, (const double*)ctx->pin_partials, nblk, cfg);
    HIPC(ctx, hipGetLastError());
    HIPC(ctx, hostwait::wait(ctx->tag, ctx->stream));
    return 0;
}

struct PinSums {
    double e2p, e2d, np, nd, rows;
    float H[36], g[6];
    double error() const { return sqrt(e2p / nd) + sqrt(e2d / nd); }     // RPI.h:742-744: both averages / nValidDepthPts
    double error_occ() const { return sqrt(e2p / np) + sqrt(e2d / nd); } // RPI.h:1314-1317, 1765-1768
};
PinSums pin_sums(const GNState& S) {
    PinSums o;
    o.e2p = S.tot[P_E2P]; o.e2d = S.tot[P_E2D]; o.np = S.tot[P_NP]; o.nd = S.tot[P_ND]; o.rows = S.tot[P_NVIS];
    int k = 0;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b, ++k) o.H[b * 6 + a] = o.H[a * 6 + b] = (float)S.tot[P_H + k];
    for (int a = 0; a < 6; ++a) o.g[a] = (float)S.tot[P_G + a];
    return o;
}

// update = -(H [+ lambda diag H])^-1 g ; pose_tmp = exp(update) * pose   (RPI.h:4355-4358, 4389-4391)
bool pin_lm_update(const float* H, const float* g, float lambda_or_neg, const float* pose, float* pose_tmp, float* update) {
    float M[36], inv[36];
    for (int k = 0; k < 36; ++k) M[k] = H[k];
    if (lambda_or_neg >= 0.f)
        for (int i = 0; i < 6; ++i) M[i * 6 + i] = H[i * 6 + i] + lambda_or_neg * H[i * 6 + i];
    if (!gn::inverse6(M, inv)) return false;
    for (int r = 0; r < 6; ++r) {
        float s = 0.f;
        for (int c = 0; c < 6; ++c) s += (-inv[c * 6 + r]) * g[c];
        update[r] = s;
    }
    double ud[6], E[16];
    for (int i = 0; i < 6; ++i) ud[i] = (double)update[i];
    gn::se3_exp(ud, E);
    float Ef[16];
    for (int k = 0; k < 16; ++k) Ef[k] = (float)E[k];
    gn::mat4_mul(Ef, pose, pose_tmp);
    return true;
}
}  // namespace

extern "C" int rgbd360_set_camera(rgbd360_ctx* ctx, float fx, float fy, float ox, float oy) {
    if (!ctx) return -1;
    if (!(fx > 0.f) || !(fy > 0.f)) return fail(ctx, -1, "focal lengths must be positive");
    ctx->cam[0] = fx; ctx->cam[1] = fy; ctx->cam[2] = ox; ctx->cam[3] = oy;
    ctx->have_cam = true;
    return 0;
}

extern "C" int rgbd360_set_index_arithmetic(rgbd360_ctx* ctx, int mode) {
    if (!ctx) return -1;
    if (mode != 0 && mode != 1) return fail(ctx, -1, "index arithmetic: 0 (device definition) or 1 (the reference's libm)");
    if (ctx->al_active) return fail(ctx, -6, "an alignment is in flight");
    ctx->index_libm = mode;
    for (Level& L : ctx->levels) L.libm = mode;
    for (SeqEngine* E : ctx->engines) E->libm = mode;
    for (rgbd360_ctx* sib : ctx->siblings) rgbd360_set_index_arithmetic(sib, mode);
    return 0;
}
extern "C" int rgbd360_get_index_arithmetic(rgbd360_ctx* ctx) { return ctx ? ctx->index_libm : -1; }

extern "C" int rgbd360_use_saliency(rgbd360_ctx* ctx, int on, float thres_saliency) {
    if (!ctx) return -1;
    if (on && !(thres_saliency >= 0.f)) return fail(ctx, -1, "thres_saliency must be >= 0");
    ctx->sal_thr = on ? thres_saliency : -1.f;
    return 0;
}

static int pin_eval_any(rgbd360_ctx* ctx, int level, const float* pose, int method, int occlusion) {
    return occlusion == 0 ? pin_eval(ctx, level, pose, method) : pin_eval_occ(ctx, level, pose, method, occlusion);
}

extern "C" int rgbd360_eval_pinhole_occ(rgbd360_ctx* ctx, int level, const float pose[16], int method, int occlusion, double err2_split[2],
                                        long long n_split[2], float H[36], float g[6], double H64[36], double g64[6], long long* n_rows) {
    int rc = pin_check(ctx, level, method);
    if (rc) return rc;
    if (!pose) return fail(ctx, -1, "null pose pointer");
    if (occlusion < 0 || occlusion > 2) return fail(ctx, -5, "occlusion must be 0, 1 or 2");
    hipSetDevice(ctx->p.device);
    if ((rc = pin_prepare_level(ctx, level)) != 0) return rc;
    if ((rc = pin_eval_any(ctx, level, pose, method, occlusion)) != 0) return rc;
    const GNState& S = *ctx->h_state;
    const PinSums P = pin_sums(S);
    if (err2_split) { err2_split[0] = P.e2p; err2_split[1] = P.e2d; }
    if (n_split) { n_split[0] = (long long)P.np; n_split[1] = (long long)P.nd; }
    if (H) memcpy(H, P.H, sizeof(P.H));
    if (g) memcpy(g, P.g, sizeof(P.g));
    if (H64 || g64) {
        int k = 0;
        for (int a = 0; a < 6; ++a)
            for (int b = a; b < 6; ++b, ++k)
                if (H64) H64[b * 6 + a] = H64[a * 6 + b] = S.tot[P_H + k];
        if (g64)
            for (int a = 0; a < 6; ++a) g64[a] = S.tot[P_G + a];
    }
    if (n_rows) *n_rows = (long long)P.rows;
    return 0;
}

extern "C" int rgbd360_eval_pinhole(rgbd360_ctx* ctx, int level, const float pose[16], int method, double err2_split[2],
                                    long long n_split[2], float H[36], float g[6], double H64[36], double g64[6], long long* n_rows) {
    return rgbd360_eval_pinhole_occ(ctx, level, pose, method, 0, err2_split, n_split, H, g, H64, g64, n_rows);
}

extern "C" int rgbd360_warp_indices_pinhole(rgbd360_ctx* ctx, int level, const float pose[16], int32_t* host_out_rc) {
    int rc = pin_check(ctx, level, 0);
    if (rc) return rc;
    if (!pose || !host_out_rc) return fail(ctx, -1, "null pointer");
    hipSetDevice(ctx->p.device);
    if ((rc = pin_prepare_level(ctx, level)) != 0) return rc;
    const Level& L = ctx->levels[level];
    int32_t* d_out = nullptr;
    HIPC(ctx, hipMalloc(&d_out, (size_t)L.n * 2 * sizeof(int32_t)));
    Pose16 P;
    memcpy(P.v, pose, sizeof(P.v));
    hipLaunchKernelGGL(k_warp_indices_pinhole, dim3((L.n + 255) / 256), dim3(256), 0, ctx->stream, pin_level_dev(L), pin_level_K(ctx, level),
                       P, d_out);
    hipError_t e = hipMemcpyAsync(host_out_rc, d_out, (size_t)L.n * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(d_out);
    HIPC(ctx, e);
    return 0;
}

extern "C" int rgbd360_align_pinhole(rgbd360_ctx* ctx, const float guess[16], int method, int occlusion, float pose_out[16],
                                     rgbd360_result* res) {
    int rc = pin_check(ctx, 0, method);
    if (rc) return rc;
    if (!guess || !pose_out) return fail(ctx, -1, "null pose pointer");
    if (occlusion < 0 || occlusion > 2) return fail(ctx, -5, "occlusion must be 0, 1 or 2");
    hipSetDevice(ctx->p.device);
    rgbd360_result R;
    memset(&R, 0, sizeof(R));
    float pose_estim[16], pose_estim_temp[16];
    memcpy(pose_estim, guess, sizeof(pose_estim));
    float H[36] = {0}, g[6] = {0};
    double last_eval = 0, last_photo = 0, last_depth = 0, temp_eval = 0, temp_photo = 0, temp_depth = 0, final_error = 0;
    bool any_iteration = false;
    int status = 0;
    PinSums P;
    float sso = 0.f;
    auto eval = [&](int level, const float* pose, double& out) -> int {
        const int e = pin_eval_any(ctx, level, pose, method, occlusion);
        if (e) return e;
        P = pin_sums(*ctx->h_state);
        out = occlusion ? P.error_occ() : P.error();
        last_eval = out;
        last_photo = sqrt(P.e2p / (occlusion ? P.np : P.nd));
        last_depth = sqrt(P.e2d / P.nd);
        return 0;
    };
    const bool pin_trace = knobs::debug("RGBD360_PIN_TRACE") != nullptr;      // debug builds: every trip's error, update, g and diag(H) on stderr
    for (int level = ctx->p.n_pyr - 1; level >= 0 && status == 0; --level) {
        if ((rc = pin_prepare_level(ctx, level)) != 0) return rc;
        float lambda = 0.01f;                 // RPI.h:4303 (double 0.01 used as a float scalar by Eigen)
        const double step = 10;
        const unsigned LM_maxIters = 1;
        int it = 0;
        const int maxIters = 10;              // RPI.h:4306-4308: the pinhole driver hard-codes its own limits
        const double tol_residual = 1e-4, tol_update = 1e-4;
        float update_pose[6] = {1, 1, 1, 1, 1, 1};
        double error = 0, new_error = 0;
        if ((rc = eval(level, pose_estim, error)) != 0) return rc;
        double diff_error = error;
        // the first pass doubles as the H,g pass of the first trip (same pose)
        PinSums at_pose = P;
        auto unorm = [&]() {
            float s2 = 0;
            for (int i = 0; i < 6; ++i) s2 += update_pose[i] * update_pose[i];
            return sqrtf(s2);
        };
        while (it < maxIters && unorm() > tol_update && diff_error > tol_residual) {
            any_iteration = true;
            temp_eval = last_eval; temp_photo = last_photo; temp_depth = last_depth;
            memcpy(H, at_pose.H, sizeof(H));          // calcHessGrad(pose_estim): the fused pass at pose_estim
            memcpy(g, at_pose.g, sizeof(g));
            if (occlusion == 2) sso = (float)(at_pose.rows / (double)ctx->levels[level].n);      // calcHessGrad_Occ2 sets SSO (RPI.h:2016)
            float M[36];
            for (int k = 0; k < 36; ++k) M[k] = H[k];
            for (int i = 0; i < 6; ++i) M[i * 6 + i] = H[i * 6 + i] + lambda * H[i * 6 + i];
            if (gn::rank6(M) != 6 || !pin_lm_update(H, g, -1.f, pose_estim, pose_estim_temp, update_pose)) {
                status = 1;                            // "The problem is ILL-POSED": relPose = pose_estim, return   RPI.h:4346-4353
                break;
            }
            PinSums cand;
            if ((rc = eval(level, pose_estim_temp, new_error)) != 0) return rc;
            cand = P;
            diff_error = error - new_error;
            if (pin_trace) {
                fprintf(stderr, "[pin trace] level %d it %d lambda %g: error %.10f -> %.10f, update", level, it, (double)lambda, error, new_error);
                for (int k = 0; k < 6; ++k) fprintf(stderr, " %.9g", (double)update_pose[k]);
                fprintf(stderr, "; g");
                for (int k = 0; k < 6; ++k) fprintf(stderr, " %.9g", (double)g[k]);
                fprintf(stderr, "; diag H");
                for (int k = 0; k < 6; ++k) fprintf(stderr, " %.9g", (double)H[k * 6 + k]);
                fprintf(stderr, "\n");
            }
            if (diff_error > 0) {
                lambda /= step;
                memcpy(pose_estim, pose_estim_temp, sizeof(pose_estim));
                error = new_error;
                it = it + 1;
                at_pose = cand;
            } else {
                unsigned LM_it = 0;
                while (LM_it < LM_maxIters && diff_error < 0) {
                    lambda = lambda * step;
                    if (!pin_lm_update(H, g, lambda, pose_estim, pose_estim_temp, update_pose)) break;
                    if ((rc = eval(level, pose_estim_temp, new_error)) != 0) return rc;
                    cand = P;
                    diff_error = error - new_error;
                    if (diff_error > 0) {
                        memcpy(pose_estim, pose_estim_temp, sizeof(pose_estim));
                        error = new_error;
                        it = it + 1;
                        at_pose = cand;
                    } else
                        LM_it = LM_it + 1;
                }
            }
        }
        if (status == 1) break;
        R.iters[level & 7] = it;
        final_error = error;
    }
    memcpy(pose_out, pose_estim, sizeof(pose_estim));
    if (status == 0 && final_error != final_error) status = 2;      // NaN: no depth-valid pixel (or PHOTO only: x / nValidDepthPts)
    R.status = status;
    R.err_final = any_iteration ? temp_eval : last_eval;             // avResidual = avResidual_temp   RPI.h:4507-4509
    R.rms_photo = any_iteration ? temp_photo : last_photo;
    R.rms_depth = any_iteration ? temp_depth : last_depth;
    if (status == 1) R.err_final = 0.0;
    R.sso = status == 1 ? 0.f : sso;
    memcpy(R.hessian, H, sizeof(H));
    memcpy(R.gradient, g, sizeof(g));
    if (res) *res = R;
    return status;
}

// test hooks (rgbd360_hip_diag.h): the schedules the parity tests compare with the default ones
extern "C" int rgbd360_debug_set_schedule(rgbd360_ctx* ctx, int fused_solve, int fused_occ) {
    if (!ctx) return -1;
    if (ctx->al_active) return fail(ctx, -6, "an alignment is in flight");
    ctx->fused_solve = fused_solve != 0;
    ctx->fused_occ = fused_occ != 0;
    return 0;
}
extern "C" int rgbd360_debug_set_sequence_route(rgbd360_ctx* ctx, int route, int max_contexts) {
    if (!ctx || route < 0 || route > 1 || max_contexts < 0 || max_contexts > 16) return -1;
    ctx->seq_route_contexts = route == 1;
    ctx->seq_route_cap = max_contexts;
    return 0;
}
extern "C" int rgbd360_debug_knobs_enabled(void) { return knobs::debug_build() ? 1 : 0; }

// the Frame360 translation unit's view of a context (f360_state.h)
F360State* rgbd360_ctx_f360(rgbd360_ctx* ctx) {
    if (!ctx->f360) ctx->f360 = f360_state_create(ctx->p.device, ctx->stream);
    return ctx->f360;
}
void rgbd360_ctx_set_error(rgbd360_ctx* ctx, const char* msg) { ctx->err = msg ? msg : ""; }

#include "multi_gpu.h"

