// rgbd360_frame360.hip -- the Frame360 side of the C ABI (include/rgbd360_hip.h): sphere / sensor clouds, the PCL normal map, planar
// regions with their moments, hulls and colour descriptors, the bilateral grid filter, the spherical stitcher (SURVEY.md 8 rows
// a13-a15, 8f rank 2).  A translation unit of its own since round 6: an experiment on the alignment kernels (rgbd360_api.hip) no longer
// rebuilds these 50 kernels and vice versa.  It sees a context only through f360_state.h: the stages' scratch lives in an F360State the
// context owns (created on the first Frame360 call, on the context's device and stream).  No CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rgbd360_hip.h"
#include "../../include/rgbd360_hip_diag.h"
#include "knobs.h"
#include "host_wait.h"
#include "device_math.h"
#include "frame360_kernels.h"
#include "pbmap_register.h"
#include "f360_state.h"

using namespace r360;

// What the stages keep between calls.  `ctx` in this file is an F360State: the member names are those the code used while it lived in
// rgbd360_ctx (stream, p.device, tag, err + the f_* / b_* scratch).
struct F360State {
    struct { int device = 0; } p;
    hipStream_t stream = nullptr;       // the owning context's stream (not owned)
    hostwait::SpinTag tag;              // pinned sequence number the stages' last kernel stores (host_wait.h); this state's own
    std::string err;                    // copied into the context's error string when an entry point returns (F360Enter)
    // Frame360 stage scratch (normals / plane segmentation), grown on demand
    size_t f360_n = 0;
    float *f_xyz = nullptr, *f_normals = nullptr, *f_dist = nullptr;
    uint8_t *f_change = nullptr, *f_hd = nullptr;
    f360::EdgeCloudSrc f_cloud_pending = {nullptr, 0, 0, 0, nullptr, nullptr, nullptr, nullptr};      // a sphere cloud the depth-edge kernel is to form (frame_planes)
    int *f_label = nullptr, *f_slot_of_root = nullptr, *f_root_of_slot = nullptr, *f_nslots = nullptr, *f_window = nullptr;
    unsigned long long *f_count = nullptr, *f_mom = nullptr;
    int* f_count_of_slot = nullptr;
    unsigned char *f_pack = nullptr, *f_pack_host = nullptr;      // packed region records: written by the device straight into pinned host memory (f_pack unused)
    f360::SlotFrame* f_frames = nullptr;                          // per region slot: centroid + in-plane basis (hull stage)
    unsigned long long* f_ext = nullptr;                          // per region slot: 256 directional extremes {ordered dot, pixel} (overflow path of the block tables)
    int* f_hull_keys = nullptr;                                   // per block of k_f360_hull_extremes: the slots of its table rows ...
    unsigned long long* f_hull_vals = nullptr;                    // ... and the rows (256 extremes each)
    int f_hull_blocks = 0;
    // colour image of the next plane calls (rgbd360_set_plane_color_image) and the per-region colour table (k_f360_colour)
    uint8_t* f_col_owned = nullptr;                               // device copy of a host image
    size_t f_col_owned_bytes = 0;
    f360::ColourImage f_col_img = {nullptr, 0, 1};
    int f_col_rows = 0, f_col_cols = 0;                           // size of the registered image
    unsigned long long *f_col = nullptr, *f_col_host = nullptr;   // [kF360MaxSlots][kColWords]: device table, pinned copy of the rows in use
    int *f_samp_off = nullptr, *f_samp_n = nullptr;               // the dominant colour's samples: where a slot's start in the pool, how many arrived
    int2* f_samp_grid = nullptr;                                  // ... and the slot's sample grid {sr, sc}
    unsigned* f_samp_pool = nullptr;                              // one entry per pixel of the largest frame seen
    size_t f_samp_pool_n = 0;
    bool f_col_ran = false;                                       // the last plane call filled f_col_host
    unsigned long long* b_sum = nullptr;                          // bilateral grid: fixed-point sums, counts, two float2 ping-pong arrays
    int* b_cnt = nullptr;
    float2 *b_a = nullptr, *b_b = nullptr;
    float4* f_models_host = nullptr;                              // pinned staging of the refinement's plane models
    unsigned* b_mm = nullptr;                                     // per-block {min, max} codes of the depth range
    unsigned* b_mm_host = nullptr;                                // ... the pair, published into pinned memory
    size_t b_cells = 0;
    float* f_tab = nullptr;
    size_t f_tab_n = 0;
    int f_tab_rows = 0, f_tab_cols = 0, f_tab_conv = -1;      // what the resident angle tables were built for
    uint8_t* f_depth_raw = nullptr;
    // measurement (rgbd360_frame_planes_stage_timing): events on `stream` at the stage boundaries of a frame_planes call -- before the
    // cloud / depth-edge kernel, behind it (row a13), behind the normal map (a14), behind the last kernel of the plane stage (a15)
    hipEvent_t f_stage_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool f_stage_timing = false, f_stage_valid = false;
    int f_planes_available = 0;     // regions that passed every filter in the last plane call (may exceed the caller's max_planes)
    int f_refine = 0;               // segmentAndRefine's refinement after `segment` (rgbd360_set_plane_refinement)
    float f_refine_dist = 0.02f;    // PlaneRefinementComparator's default distance threshold
    int f_refine_changed = 0, f_refine_sweeps = 0;      // pixels relabelled / Jacobi sweeps of the last call
    float4* f_models = nullptr;     // per slot {a, b, c, d} of the planes `segment` produced (x = NaN: no plane)
    int* f_flags_host = nullptr;    // pinned, device-visible: per-sweep "something changed" flags + the relabelled-pixel counter
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

int fail(F360State* ctx, int code, const char* msg) {
    ctx->err = msg;
    return code;
}
dim3 grid2d(int rows, int cols, int bx = 256) { return dim3((cols + bx - 1) / bx, rows, 1); }

// Every entry point: the context's F360State (created on first use) as `ctx`; whatever error text the call leaves goes into the context
// (rgbd360_last_error) on every return path.
struct F360Enter {
    rgbd360_ctx* c;
    F360State* s;
    explicit F360Enter(rgbd360_ctx* c_) : c(c_), s(c_ ? rgbd360_ctx_f360(c_) : nullptr) {
        if (s) s->err.clear();
    }
    ~F360Enter() {
        if (s && !s->err.empty()) rgbd360_ctx_set_error(c, s->err.c_str());
    }
};
#define F360_ENTER(c_)                                                    \
    F360Enter enter_((c_));                                               \
    F360State* ctx = enter_.s;                                            \
    if (!ctx) return (c_) ? -103 : -1
}  // namespace

F360State* f360_state_create(int device, hipStream_t stream) {
    F360State* s = new F360State();
    s->p.device = device;
    s->stream = stream;
    if (hostwait::spin_tag_init(&s->tag) != hipSuccess) {
        delete s;
        return nullptr;
    }
    return s;
}
void f360_state_destroy(F360State* ctx) {
    if (!ctx) return;
    hipFree(ctx->f_frames); hipFree(ctx->f_ext); hipFree(ctx->f_hull_keys); hipFree(ctx->f_hull_vals);
    hipFree(ctx->f_xyz); hipFree(ctx->f_normals); hipFree(ctx->f_dist);
    hipFree(ctx->f_change); hipFree(ctx->f_hd); hipFree(ctx->f_label); hipFree(ctx->f_count); hipFree(ctx->f_slot_of_root);
    hipFree(ctx->f_root_of_slot); hipFree(ctx->f_nslots); hipFree(ctx->f_window); hipFree(ctx->f_mom);
    hipFree(ctx->f_count_of_slot); hipFree(ctx->f_depth_raw); hipFree(ctx->f_pack);
    if (ctx->f_pack_host) hipHostFree(ctx->f_pack_host);
    hipFree(ctx->f_col_owned); hipFree(ctx->f_col); hipFree(ctx->f_samp_off); hipFree(ctx->f_samp_n); hipFree(ctx->f_samp_grid); hipFree(ctx->f_samp_pool);
    ctx->f_samp_off = ctx->f_samp_n = nullptr; ctx->f_samp_grid = nullptr; ctx->f_samp_pool = nullptr; ctx->f_samp_pool_n = 0;
    if (ctx->f_col_host) hipHostFree(ctx->f_col_host);
    ctx->f_col_owned = nullptr; ctx->f_col = nullptr; ctx->f_col_host = nullptr; ctx->f_col_owned_bytes = 0;
    hipFree(ctx->f_models);
    if (ctx->f_flags_host) hipHostFree(ctx->f_flags_host);
    hipFree(ctx->b_sum); hipFree(ctx->b_a); hipFree(ctx->b_b); hipFree(ctx->b_mm); if (ctx->b_mm_host) hipHostFree(ctx->b_mm_host); if (ctx->f_models_host) hipHostFree(ctx->f_models_host);
    for (hipEvent_t e : ctx->f_stage_ev)
        if (e) hipEventDestroy(e);
    hostwait::spin_tag_free(&ctx->tag);
    delete ctx;
}

// ---------------------------------------------------------------------------------------------------------
// Frame360 stages: normal map (row a14) and planar regions + inlier moments (row a15)
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr int kF360MaxSlots = 4096;

int f360_ensure(F360State* ctx, size_t n) {
    if (ctx->f360_n >= n) return 0;
    hipFree(ctx->f_xyz); hipFree(ctx->f_normals); hipFree(ctx->f_dist);
    hipFree(ctx->f_change); hipFree(ctx->f_hd); hipFree(ctx->f_label); hipFree(ctx->f_count); hipFree(ctx->f_slot_of_root);
    hipFree(ctx->f_root_of_slot); hipFree(ctx->f_nslots); hipFree(ctx->f_window); hipFree(ctx->f_mom);
    hipFree(ctx->f_count_of_slot); hipFree(ctx->f_depth_raw); hipFree(ctx->f_pack); hipFree(ctx->f_frames); hipFree(ctx->f_ext); hipFree(ctx->f_hull_keys); hipFree(ctx->f_hull_vals);
    if (ctx->f_pack_host) hipHostFree(ctx->f_pack_host);
    ctx->f_frames = nullptr; ctx->f_ext = nullptr; ctx->f_hull_keys = nullptr; ctx->f_hull_vals = nullptr;
    ctx->f_xyz = ctx->f_normals = ctx->f_dist = nullptr;         // a failed allocation below must not leave freed pointers behind
    ctx->f_change = nullptr; ctx->f_hd = nullptr; ctx->f_label = nullptr; ctx->f_count = nullptr; ctx->f_slot_of_root = nullptr;
    ctx->f_root_of_slot = nullptr; ctx->f_nslots = nullptr; ctx->f_window = nullptr; ctx->f_mom = nullptr;
    ctx->f_count_of_slot = nullptr; ctx->f_depth_raw = nullptr; ctx->f_pack = nullptr; ctx->f_pack_host = nullptr;
    ctx->f360_n = 0;
    HIPC(ctx, hipMalloc(&ctx->f_xyz, n * 3 * sizeof(float)));
    HIPC(ctx, hipMalloc(&ctx->f_normals, n * 3 * sizeof(float)));
    HIPC(ctx, hipMalloc(&ctx->f_dist, n * sizeof(float)));
    HIPC(ctx, hipMalloc(&ctx->f_change, n));
    HIPC(ctx, hipMalloc(&ctx->f_hd, 3 * n + 64));      // depth-change bit mask: rows x ceil(cols / 64) words <= n/8 + 8 rows bytes, cols >= 3
    HIPC(ctx, hipMalloc(&ctx->f_label, n * sizeof(int)));
    HIPC(ctx, hipMalloc(&ctx->f_count, n * sizeof(unsigned long long)));
    HIPC(ctx, hipMalloc(&ctx->f_slot_of_root, n * sizeof(int)));
    HIPC(ctx, hipMalloc(&ctx->f_window, n * sizeof(int)));
    HIPC(ctx, hipMalloc(&ctx->f_root_of_slot, kF360MaxSlots * sizeof(int)));
    HIPC(ctx, hipMalloc(&ctx->f_nslots, sizeof(int)));
    HIPC(ctx, hipMalloc(&ctx->f_mom, (size_t)f360::kMomReplicas * kF360MaxSlots * 9 * sizeof(unsigned long long)));
    HIPC(ctx, hipMalloc(&ctx->f_count_of_slot, kF360MaxSlots * sizeof(int)));
    // pinned: header, one moment record per slot, one hull record per slot behind them
    const size_t pack_bytes = f360::kF360PackHeader + (size_t)kF360MaxSlots * (sizeof(f360::F360SlotRecord) + sizeof(f360::F360HullRecord));
    HIPC(ctx, hipHostMalloc(&ctx->f_pack_host, pack_bytes, hostwait::kPublishedFlags));
    HIPC(ctx, hipMalloc(&ctx->f_frames, (size_t)kF360MaxSlots * sizeof(f360::SlotFrame)));
    HIPC(ctx, hipMalloc(&ctx->f_ext, (size_t)kF360MaxSlots * f360::kHullPhases * f360::kHullDirs * sizeof(unsigned long long)));
    ctx->f_hull_blocks = (int)((n + (size_t)f360::kHullBlock * f360::kHullChunks - 1) / ((size_t)f360::kHullBlock * f360::kHullChunks));
    HIPC(ctx, hipMalloc(&ctx->f_hull_keys, (size_t)ctx->f_hull_blocks * f360::kHullHash * sizeof(int)));
    HIPC(ctx, hipMalloc(&ctx->f_hull_vals, (size_t)ctx->f_hull_blocks * f360::kHullHash * f360::kHullDirs * sizeof(unsigned long long)));
    HIPC(ctx, hipMalloc(&ctx->f_depth_raw, n * 4));
    ctx->f360_n = n;
    return 0;
}

// distance to the nearest depth change of the organised cloud in ctx->f_xyz -> ctx->f_dist (device); f_hd holds the bit mask
static void launch_distance_map(F360State* ctx, int rows, int cols, float max_depth_change_factor, int depth_mode, unsigned* clear_words = nullptr,
                                int n_clear = 0) {
    using namespace f360;
    const int pitch = (cols + 63) / 64;
    unsigned long long* bits = reinterpret_cast<unsigned long long*>(ctx->f_hd);
    const dim3 ge((cols + kEdgeTW - 1) / kEdgeTW, (rows + kEdgeTH - 1) / kEdgeTH);
    if (ctx->f_cloud_pending.depth) {       // rgbd360_frame_planes: the cloud has not been formed yet -- this kernel does it on the way
        const EdgeCloudSrc& cs = ctx->f_cloud_pending;
        const int spec = (cs.convention >= 0 && cs.convention <= 2 && (cs.depth_type | 1) == 1 && (depth_mode | 1) == 1) ? cs.convention * 4 + cs.depth_type * 2 + depth_mode : -1;
#define EDGE_CLOUD(S) hipLaunchKernelGGL((k_f360_edge_bits<true, S>), ge, dim3(kEdgeTW), 0, ctx->stream, ctx->f_xyz, rows, cols, max_depth_change_factor, depth_mode, \
                                         pitch, bits, ctx->f_cloud_pending, ctx->f_xyz)
        switch (spec) {
            case 0: EDGE_CLOUD(0); break;
            case 1: EDGE_CLOUD(1); break;
            case 2: EDGE_CLOUD(2); break;
            case 3: EDGE_CLOUD(3); break;
            case 4: EDGE_CLOUD(4); break;
            case 5: EDGE_CLOUD(5); break;
            case 6: EDGE_CLOUD(6); break;
            case 7: EDGE_CLOUD(7); break;
            case 8: EDGE_CLOUD(8); break;
            case 9: EDGE_CLOUD(9); break;
            case 10: EDGE_CLOUD(10); break;
            case 11: EDGE_CLOUD(11); break;
            default: EDGE_CLOUD(-1); break;
        }
#undef EDGE_CLOUD
        ctx->f_cloud_pending.depth = nullptr;
    } else {
        hipLaunchKernelGGL((k_f360_edge_bits<false>), ge, dim3(kEdgeTW), 0, ctx->stream, ctx->f_xyz, rows, cols, max_depth_change_factor, depth_mode,
                           pitch, bits, ctx->f_cloud_pending, ctx->f_xyz);
    }
    if (ctx->f_stage_timing) hipEventRecord(ctx->f_stage_ev[1], ctx->stream);
    hipLaunchKernelGGL(k_f360_distmap, dim3(pitch, (rows + kDistTH - 1) / kDistTH), dim3(kDistThreads), 0, ctx->stream, bits, rows, cols,
                       pitch, ctx->f_dist, clear_words, n_clear);
}

// normals of the organised cloud in ctx->f_xyz -> ctx->f_normals (device)
int f360_normals_dev(F360State* ctx, int rows, int cols, float max_depth_change_factor, float smoothing_size, int depth_mode) {
    using namespace f360;
    if (smoothing_size < 1.f || smoothing_size + 2.5f > (float)kF360R)
        return fail(ctx, -1, "normal_smoothing_size out of range (the distance map is truncated at 12 px)");
    const dim3 gt((cols + kNT_W - 1) / kNT_W, (rows + kNT_H - 1) / kNT_H);
    // Register sweep for the pixels whose window is int(smoothing_size) squared (nearly all of them), the tiled integral-image kernel
    // for the 32 x 16 tiles the sweep lists (depth edges, far points).  Claim flags + tile list: the first bytes of f_change, which the
    // distance map does not touch (the plane stage rewrites it); the distance-map kernel clears them on its way.
    const int R = (int)smoothing_size;
    const bool use_sweep = R >= 3 && R <= 10 && ((size_t)gt.x * gt.y * 2 + 2) * sizeof(unsigned) <= (size_t)rows * cols &&
                           (size_t)rows * cols * 12 < ((size_t)1 << 31);      // the sweep addresses its rows with 32-bit buffer offsets
    const int n_tiles = (int)(gt.x * gt.y);
    unsigned *flags = nullptr, *list = nullptr;            // {claimed flag per tile}, {count, tile ids ...}
    if (use_sweep) {
        flags = reinterpret_cast<unsigned*>(ctx->f_change);
        list = flags + n_tiles;
    }
    launch_distance_map(ctx, rows, cols, max_depth_change_factor, depth_mode, flags, use_sweep ? n_tiles + 1 : 0);
    if (use_sweep) {
        // rows per wave: 3 waves per SIMD stay resident (146 VGPRs: 3072 on the chip); a wave costs R - 1 warm-up rows + its rows, and
        // the kernel ends with the most loaded SIMD, so the wave count is kept within one residency round.  16 rows per wave is the
        // measured optimum at 2048 x 1024 (28.6 us; 8: 28.5, 24: 31.6, 32: 39.4), 16-32 at 4096 x 2048 (85-88 us; 48: 101) --
        // tools/normals_seg_sweep.sh
        const int OW = 63 - R;
        const int strips = (cols + OW - 1) / OW;
        const int rounds = std::max(1, (int)lround((double)strips * rows / 32.0 / 3072.0));
        const int segs = std::max(1, 3072 * rounds / strips);
        // after R - 1 warm-up rows a wave sweeps whole passes of R output rows (its unrolled loop body)
        int seg = std::max(2 * R, (rows + segs - 1) / segs);
        seg = (seg + R - 1) / R * R;
        if (const char* e = knobs::debug("RGBD360_SWEEP_SEG")) {      // tuning knob
            const int v = atoi(e);
            if (v >= 4 && v <= 4096) seg = v;
        }
        const int units = strips * ((rows + seg - 1) / seg);
        const dim3 gs((units + kSweepWaves - 1) / kSweepWaves), bs(64 * kSweepWaves);
        // (no window plane: nothing downstream reads the per-pixel window size -- it was a 4 B/px store of both normal-map kernels,
        // 33 MB at 4096 x 2048, kept from the days the two kernels were compared through it; round 6)
#define SWEEP(RR) hipLaunchKernelGGL((k_f360_normals_sweep<RR>), gs, bs, 0, ctx->stream, ctx->f_xyz, ctx->f_dist, rows, cols, smoothing_size, depth_mode, seg, ctx->f_normals, (int*)nullptr, flags, list, (int)gt.x)
        switch (R) {
            case 3: SWEEP(3); break;
            case 4: SWEEP(4); break;
            case 5: SWEEP(5); break;
            case 6: SWEEP(6); break;
            case 7: SWEEP(7); break;
            case 8: SWEEP(8); break;
            case 9: SWEEP(9); break;
            default: SWEEP(10); break;
        }
#undef SWEEP
    }
    // two tiles fit a CU: 512 blocks walk the sweep's list (every tile of the frame when there was no sweep)
    hipLaunchKernelGGL(k_f360_normals_tiled, dim3(list ? std::min(n_tiles, 512) : n_tiles), dim3(kNT_THREADS), 0, ctx->stream, ctx->f_xyz,
                       ctx->f_dist, rows, cols, smoothing_size, depth_mode, ctx->f_normals, (int*)nullptr, (const unsigned*)list, (int)gt.x,
                       n_tiles);
    if (ctx->f_stage_timing) hipEventRecord(ctx->f_stage_ev[2], ctx->stream);
    HIPC(ctx, hipGetLastError());
    return 0;
}

// pcl::FastBilateralFilter on the organised cloud in ctx->f_xyz (device), z filtered in place
int f360_bilateral_dev(F360State* ctx, int rows, int cols, float sigma_s, float sigma_r) {
    using namespace f360;
    if (!(sigma_s > 0.f) || !(sigma_r > 0.f)) return fail(ctx, -1, "sigma_s and sigma_r must be positive");
    const int n = rows * cols;
    constexpr int kMmBlocks = 64;
    if (!ctx->b_mm) HIPC(ctx, hipMalloc(&ctx->b_mm, 2 * kMmBlocks * sizeof(unsigned)));
    if (!ctx->b_mm_host) HIPC(ctx, hipHostMalloc((void**)&ctx->b_mm_host, 2 * sizeof(unsigned), hostwait::kPublishedFlags));
    // the depth range: per-block pairs, folded and published by a one-wave kernel; the host spins on the tag (no memset, copy or stream synchronise)
    const int mm_blocks = std::min(kMmBlocks, (n + 4 * kBilatMmThreads - 1) / (4 * kBilatMmThreads));
    hipLaunchKernelGGL(k_bilat_minmax, dim3(mm_blocks), dim3(kBilatMmThreads), 0, ctx->stream, ctx->f_xyz, n, ctx->b_mm);
    hipLaunchKernelGGL(k_bilat_minmax_publish, dim3(1), dim3(64), 0, ctx->stream, ctx->b_mm, mm_blocks, ctx->b_mm_host, ctx->tag.h, ++ctx->tag.seq);
    HIPC(ctx, hipGetLastError());
    HIPC(ctx, hostwait::wait(ctx->tag, ctx->stream));
    const unsigned mm[2] = {reinterpret_cast<const volatile unsigned*>(ctx->b_mm_host)[0], reinterpret_cast<const volatile unsigned*>(ctx->b_mm_host)[1]};
    if (mm[0] > mm[1]) return 0;                                   // no finite z: the cloud stays as it is
    auto decode = [](unsigned e) {
        const unsigned u = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
        float v;
        memcpy(&v, &u, sizeof(v));
        return v;
    };
    BilatGrid g;
    g.sigma_s = sigma_s; g.sigma_r = sigma_r;
    g.base_min = decode(mm[0]); g.base_max = decode(mm[1]);
    const float base_delta = g.base_max - g.base_min;
    g.nx = (int)((float)(cols - 1) / sigma_s) + 1 + 2 * kBilatPadXY;
    g.ny = (int)((float)(rows - 1) / sigma_s) + 1 + 2 * kBilatPadXY;
    const double nz = (double)(base_delta / sigma_r) + 1 + 2 * kBilatPadZ;
    const double cells_d = (double)g.nx * g.ny * nz;
    if (!(cells_d < 64e6)) return fail(ctx, -1, "bilateral grid too large (depth range / sigma_r)");
    g.nz = (int)(base_delta / sigma_r) + 1 + 2 * kBilatPadZ;
    const size_t cells = (size_t)g.nx * g.ny * g.nz;
    if (ctx->b_cells < cells) {
        hipFree(ctx->b_sum); hipFree(ctx->b_a); hipFree(ctx->b_b);
        ctx->b_sum = nullptr; ctx->b_cnt = nullptr; ctx->b_a = ctx->b_b = nullptr; ctx->b_cells = 0;
        HIPC(ctx, hipMalloc(&ctx->b_sum, cells * (sizeof(unsigned long long) + sizeof(int))));      // sums, then counts: one allocation, one memset
        ctx->b_cnt = reinterpret_cast<int*>(ctx->b_sum + cells);
        HIPC(ctx, hipMalloc(&ctx->b_a, cells * sizeof(float2)));
        HIPC(ctx, hipMalloc(&ctx->b_b, cells * sizeof(float2)));
        ctx->b_cells = cells;
    }
    ctx->b_cnt = reinterpret_cast<int*>(ctx->b_sum + cells);      // (behind THIS call's cells: the buffer may be larger)
    HIPC(ctx, hipMemsetAsync(ctx->b_sum, 0, cells * (sizeof(unsigned long long) + sizeof(int)), ctx->stream));
    const dim3 gp((n + 255) / 256), gc((unsigned)((cells + 255) / 256)), b(256);
    hipLaunchKernelGGL(k_bilat_scatter, gp, b, 0, ctx->stream, ctx->f_xyz, rows, cols, g, ctx->b_sum, ctx->b_cnt);
    hipLaunchKernelGGL(k_bilat_init, gc, b, 0, ctx->stream, ctx->b_sum, ctx->b_cnt, cells, ctx->b_a, ctx->b_b);
    float2 *data = ctx->b_a, *buffer = ctx->b_b;
    const int offs[3] = {g.ny * g.nz, g.nz, 1};
    for (int dim = 0; dim < 3; ++dim)
        for (int it = 0; it < 2; ++it) {
            std::swap(data, buffer);
            hipLaunchKernelGGL(k_bilat_blur, gc, b, 0, ctx->stream, buffer, data, g, offs[dim]);
        }
    hipLaunchKernelGGL(k_bilat_interp, gp, b, 0, ctx->stream, ctx->f_xyz, rows, cols, g, data);
    HIPC(ctx, hipGetLastError());
    return 0;
}

// Convex hull of the directional extremes of a region (<= kHullRecPts points), its area (shoelace) and mass centre -- the roles of
// mrpt::pbmap::Plane::calcConvexHull / computeMassCenterAndArea (Frame360.h:1025-1031) in the region's own in-plane frame.
struct HullStats {
    int n = 0;                  // hull vertices
    double area = 0, cu = 0, cv = 0;
    int np = 0;                 // vertices kept for the plane record (<= RGBD360_HULL_MAX), counter-clockwise in (u, v)
    double pu[RGBD360_HULL_MAX], pv[RGBD360_HULL_MAX];
};
struct HullPt {
    double x, y;                // (no constructor: the work arrays below are not cleared -- 48 KB per plane when they were std::pairs)
    bool operator<(const HullPt& o) const { return x < o.x || (x == o.x && y < o.y); }
    bool operator==(const HullPt& o) const { return x == o.x && y == o.y; }
};
HullStats hull_stats(const float (*uv)[2], int K) {
    // The points arrive roughly in boundary order (direction order), but the four direction sets each saw their own sample of the
    // region's pixels, so neighbours may be swapped along an edge: Andrew's monotone chain on the sorted points, which assumes nothing.
    HullPt p[f360::kHullRecPts];
    int n = 0;
    for (int k = 0; k < K && k < f360::kHullRecPts; ++k) {
        const float a = uv[k][0], b = uv[k][1];            // (one read each of the pinned record)
        if (std::isfinite(a) && std::isfinite(b)) { p[n].x = a; p[n].y = b; ++n; }
    }
    std::sort(p, p + n);
    n = (int)(std::unique(p, p + n) - p);
    HullStats h;
    if (n < 3) return h;
    auto cross = [](const HullPt& o, const HullPt& a, const HullPt& b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); };
    HullPt H[2 * f360::kHullRecPts + 2];
    int m = 0;
    for (int i = 0; i < n; ++i) {                          // lower hull
        while (m >= 2 && cross(H[m - 2], H[m - 1], p[i]) <= 0) --m;
        H[m++] = p[i];
    }
    for (int i = n - 2, lo = m + 1; i >= 0; --i) {         // upper hull
        while (m >= lo && cross(H[m - 2], H[m - 1], p[i]) <= 0) --m;
        H[m++] = p[i];
    }
    --m;                                                   // the first point again
    if (m < 3) return h;
    double a2 = 0, cu = 0, cv = 0;
    for (int i = 0; i < m; ++i) {
        const HullPt &a = H[i], &b = H[(i + 1) % m];
        const double cr = a.x * b.y - a.y * b.x;
        a2 += cr;
        cu += (a.x + b.x) * cr;
        cv += (a.y + b.y) * cr;
    }
    if (!(fabs(a2) > 0)) return h;
    h.n = m;
    h.area = fabs(a2) / 2;
    h.cu = cu / (3 * a2);
    h.cv = cv / (3 * a2);
    // the polygon of the record: the hull itself, or its extreme vertices in RGBD360_HULL_MAX evenly spaced directions (Andrew's chain
    // leaves the vertices counter-clockwise, so the picks come out in hull order)
    if (m <= RGBD360_HULL_MAX) {
        for (int i = 0; i < m; ++i) { h.pu[i] = H[i].x; h.pv[i] = H[i].y; }
        h.np = m;
    } else {
        // (the directions turn counter-clockwise like the vertices: the extreme vertex only ever moves forward -- one full scan for the
        // first direction, then a walk; with a scan per direction this loop was 4 us per plane, 0.19 ms of a 4096 x 2048 frame's call)
        static const struct Dirs {
            double c[RGBD360_HULL_MAX], s[RGBD360_HULL_MAX];
            Dirs() { for (int k = 0; k < RGBD360_HULL_MAX; ++k) { const double th = 2.0 * 3.14159265358979323846 * k / RGBD360_HULL_MAX; c[k] = cos(th); s[k] = sin(th); } }
        } D;
        auto proj = [&](int i, int k) { return ((double)H[i].x - h.cu) * D.c[k] + ((double)H[i].y - h.cv) * D.s[k]; };
        int best = 0;
        for (int i = 1; i < m; ++i)
            if (proj(i, 0) > proj(best, 0)) best = i;
        int last = -1, first = -1;
        for (int k = 0; k < RGBD360_HULL_MAX; ++k) {
            for (int steps = 0; steps < m && proj((best + 1) % m, k) > proj(best, k); ++steps) best = (best + 1) % m;
            if (best == last || best == first) continue;
            if (first < 0) first = best;
            last = best;
            h.pu[h.np] = H[best].x; h.pv[h.np] = H[best].y; ++h.np;
        }
    }
    return h;
}
// area / centre of plane P from the hull record of its slot (pinned, written by k_f360_hull_pack); the moment rectangle stays in
// area_moment, and is the fallback when a region has fewer than three extreme points
void apply_hull(rgbd360_plane& P, const f360::F360HullRecord& R) {
    const HullStats h = hull_stats(R.uv, std::min(std::max(R.n, 0), f360::kHullRecPts));
    P.hull_points = h.n;
    P.hull_n = 0;
    if (h.n >= 3) {
        P.area = (float)h.area;
        for (int k = 0; k < 3; ++k) P.center_hull[k] = (float)((double)R.c[k] + h.cu * (double)R.e1[k] + h.cv * (double)R.e2[k]);
        // counter-clockwise seen from the side the record's normal points to: (e1, e2, e1 x e2) is right-handed, so the (u, v) order is
        // counter-clockwise about e1 x e2 -- reversed when the normal points the other way
        const double e3[3] = {(double)R.e1[1] * R.e2[2] - (double)R.e1[2] * R.e2[1], (double)R.e1[2] * R.e2[0] - (double)R.e1[0] * R.e2[2],
                              (double)R.e1[0] * R.e2[1] - (double)R.e1[1] * R.e2[0]};
        const bool flip = e3[0] * P.normal[0] + e3[1] * P.normal[1] + e3[2] * P.normal[2] < 0;
        P.hull_n = h.np;
        for (int i = 0; i < h.np; ++i) {
            const int src = flip ? h.np - 1 - i : i;
            for (int k = 0; k < 3; ++k) P.hull[i][k] = (float)((double)R.c[k] + h.pu[src] * (double)R.e1[k] + h.pv[src] * (double)R.e2[k]);
        }
    } else {
        P.area = P.area_moment;
        for (int k = 0; k < 3; ++k) P.center_hull[k] = P.centroid[k];
    }
}
const f360::F360HullRecord* hull_records(const F360State* ctx) {
    return reinterpret_cast<const f360::F360HullRecord*>(ctx->f_pack_host + f360::kF360PackHeader + (size_t)kF360MaxSlots * sizeof(f360::F360SlotRecord));
}
// the extremes of the CURRENT labels (ctx->f_label) against the frames of the slots, packed for the host; enqueued on the stream
int launch_hull(F360State* ctx, int rows, int cols, bool clear_first) {
    using namespace f360;
    const int n = rows * cols;
    if (clear_first) hipLaunchKernelGGL(k_f360_hull_clear, dim3((kF360MaxSlots * kHullPhases * kHullDirs + 255) / 256), dim3(256), 0, ctx->stream, ctx->f_nslots, kF360MaxSlots, ctx->f_ext);
    const int nblk = (n + kHullBlock * kHullChunks - 1) / (kHullBlock * kHullChunks);      // <= ctx->f_hull_blocks (sized for the context's largest frame)
    static const int n_cus = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    if (nblk > n_cus)        // more blocks than CUs: the two-per-CU build of the kernel
        hipLaunchKernelGGL(k_f360_hull_extremes_two, dim3(nblk), dim3(kHullBlock), 0, ctx->stream, ctx->f_xyz, ctx->f_label, ctx->f_slot_of_root, rows, cols,
                           ctx->f_frames, ctx->f_ext, ctx->f_hull_keys, ctx->f_hull_vals, kHullFramesLds);
    else
        hipLaunchKernelGGL(k_f360_hull_extremes, dim3(nblk), dim3(kHullBlock), 0, ctx->stream, ctx->f_xyz, ctx->f_label, ctx->f_slot_of_root, rows, cols,
                           ctx->f_frames, ctx->f_ext, ctx->f_hull_keys, ctx->f_hull_vals, kHullFramesLds);
    hipLaunchKernelGGL(k_f360_hull_merge, dim3(64, kHullMergeSplit), dim3(kHullDirs), 0, ctx->stream, ctx->f_nslots, kF360MaxSlots, ctx->f_hull_keys, ctx->f_hull_vals,
                       nblk, ctx->f_ext);
    hipLaunchKernelGGL(k_f360_hull_pack, dim3(256), dim3(kHullDirs), 0, ctx->stream, ctx->f_xyz, ctx->f_frames, ctx->f_ext, ctx->f_nslots, kF360MaxSlots,
                       const_cast<F360HullRecord*>(hull_records(ctx)));
    return 0;
}

// eigenpairs of a symmetric 3x3 in ascending order (cyclic Jacobi, float64) -- pcl::eigen33's role for the smallest one
void sorted_eigen3(const double C[3][3], double evals[3], double evecs[3][3]) {    // evecs[k] = eigenvector of evals[k]
    double ev[3], V[3][3];
    pbm::jacobi3(C, ev, V);
    int o[3] = {0, 1, 2};
    std::stable_sort(o, o + 3, [&](int a, int c) { return ev[a] < ev[c]; });
    for (int k = 0; k < 3; ++k) {
        evals[k] = ev[o[k]];
        for (int i = 0; i < 3; ++i) evecs[k][i] = V[i][o[k]];
    }
}

// Colour descriptors of the regions in their slots (k_f360_colour over the CURRENT labels), enqueued on the stream: the table rows in
// use land in pinned host memory.  Only when a colour image of this cloud's geometry is registered.
bool launch_colour(F360State* ctx, int rows, int cols) {
    using namespace f360;
    ctx->f_col_ran = false;
    const ColourImage& im = ctx->f_col_img;
    if (!im.rgb || im.sub < 1 || ctx->f_col_rows / im.sub != rows || ctx->f_col_cols / im.sub != cols) return false;
    const size_t bytes = (size_t)kF360MaxSlots * kColWords * sizeof(unsigned long long);
    if (!ctx->f_col && hipMalloc(&ctx->f_col, bytes) != hipSuccess) return false;
    if (!ctx->f_col_host && hipHostMalloc((void**)&ctx->f_col_host, bytes, hostwait::kPublishedFlags) != hipSuccess) return false;
    const int n = rows * cols;
    // the dominant colour's sample pool (one entry per pixel bounds the sum of min(count, kModeCap) over the regions)
    if (!ctx->f_samp_off && hipMalloc(&ctx->f_samp_off, kF360MaxSlots * sizeof(int)) != hipSuccess) return false;
    if (!ctx->f_samp_n && hipMalloc(&ctx->f_samp_n, kF360MaxSlots * sizeof(int)) != hipSuccess) return false;
    if (!ctx->f_samp_grid && hipMalloc(&ctx->f_samp_grid, kF360MaxSlots * sizeof(int2)) != hipSuccess) return false;
    if (ctx->f_samp_pool_n < (size_t)n) {
        hipFree(ctx->f_samp_pool);
        ctx->f_samp_pool = nullptr;
        ctx->f_samp_pool_n = 0;
        if (hipMalloc(&ctx->f_samp_pool, (size_t)n * sizeof(unsigned)) != hipSuccess) return false;
        ctx->f_samp_pool_n = (size_t)n;
    }
    const ColourSamples smp = {ctx->f_count_of_slot, ctx->f_samp_off, ctx->f_samp_n, ctx->f_samp_grid, ctx->f_samp_pool};
    hipLaunchKernelGGL(k_f360_colour_offsets, dim3(1), dim3(1024), 0, ctx->stream, ctx->f_nslots, kF360MaxSlots, ctx->f_count_of_slot, ctx->f_samp_off, ctx->f_samp_n,
                       ctx->f_samp_grid, ctx->f_col);
    hipLaunchKernelGGL(k_f360_colour, dim3((n + kAggThreads * kColPerThread - 1) / (kAggThreads * kColPerThread)), dim3(kAggThreads), 0, ctx->stream,
                       ctx->f_label, ctx->f_slot_of_root, rows, cols, im, ctx->f_col, smp);
    hipLaunchKernelGGL(k_f360_colour_mode, dim3(256), dim3(kModeThreads), 0, ctx->stream, ctx->f_nslots, kF360MaxSlots, smp, ctx->f_col, ctx->f_col_host);
    ctx->f_col_ran = true;
    return true;
}
// sums of a slot's row -> the plane's colour fields (rgbd360_hip.h); the record stays colourless when no colour pass ran
void apply_colour(const F360State* ctx, rgbd360_plane& P, int slot) {
    P.color_count = 0;
    for (int k = 0; k < 3; ++k) P.color_nrgb[k] = P.color_dev[k] = 0.f;
    P.intensity = 0.f;
    for (int k = 0; k < 74; ++k) P.hist_h[k] = 0.f;
    P.color_mode_count = 0;
    for (int k = 0; k < 3; ++k) P.color_mode[k] = 0.f;
    P.intensity_mode = P.color_concentration = 0.f;
    if (!ctx->f_col_ran) return;
    const volatile unsigned long long* w = ctx->f_col_host + (size_t)slot * f360::kColWords;
    const double n = (double)w[7];
    unsigned long long total = 0;
    for (int k = 0; k < f360::kColBins; ++k) total += w[f360::kColSums + k];
    if (total == 0) return;
    P.color_count = (int)w[7];
    if (n > 0) {
        for (int k = 0; k < 3; ++k) {
            const double m = (double)w[k] / n / 65536.0;
            const double var = (double)w[3 + k] / n / (65536.0 * 65536.0) - m * m;
            P.color_nrgb[k] = (float)m;
            P.color_dev[k] = (float)sqrt(std::max(var, 0.0));
        }
        P.intensity = (float)((double)w[6] / n);
    }
    for (int k = 0; k < f360::kColBins; ++k) P.hist_h[k] = (float)((double)w[f360::kColSums + k] / (double)total);
    const volatile unsigned long long* md = w + f360::kColSums + f360::kColBins;      // k_f360_colour_mode: N, kept, mode q (3), sum S, iterations, threshold^2
    if (md[0] > 0 && md[1] > 0) {
        P.color_mode_count = (int)md[0];
        for (int k = 0; k < 3; ++k) P.color_mode[k] = (float)((double)md[2 + k] / 65536.0);
        P.intensity_mode = (float)((double)md[5] / (double)md[1]);
        P.color_concentration = (float)((double)md[1] / (double)md[0]);
    }
}

// regions of (ctx->f_xyz, ctx->f_normals) -> labels (device ctx->f_label) + plane list (host)
// segmentAndRefine's refinement on the device (frame360_kernels.h k_f360_refine_tile): block-Jacobi steps of the two raster passes
// until nothing changes, then the grown inliers are added to their planes' integer sums and the extent descriptors recomputed.
int f360_refine_dev(F360State* ctx, int rows, int cols, int nslots, std::vector<rgbd360_plane>& planes, const std::vector<int>& plane_slot) {
    using namespace f360;
    const int n = rows * cols;
    if (!ctx->f_models) HIPC(ctx, hipMalloc(&ctx->f_models, (kF360MaxSlots + 1) * sizeof(float4)));      // + one slot: the relabelled-pixel counter
    int* d_changed = reinterpret_cast<int*>(ctx->f_models + kF360MaxSlots);     // (device memory: 166 k atomics into pinned host memory took 8 ms)
    int* d_activity = d_changed + 1;                                            // bumped by every relaxation step that changed a label
    constexpr int kFlags = 64;
    if (!ctx->f_flags_host) HIPC(ctx, hipHostMalloc((void**)&ctx->f_flags_host, (kFlags + 1) * sizeof(int), hostwait::kPublishedFlags));
    // (the plane models go up from a pinned buffer: a copy out of pageable memory is staged by the runtime)
    if (!ctx->f_models_host) HIPC(ctx, hipHostMalloc((void**)&ctx->f_models_host, kF360MaxSlots * sizeof(float4), 0));
    float4* models = ctx->f_models_host;
    for (int s2 = 0; s2 < nslots; ++s2) models[s2] = make_float4(NAN, 0.f, 0.f, 0.f);
    for (size_t k = 0; k < planes.size(); ++k)
        models[plane_slot[k]] = make_float4(planes[k].normal[0], planes[k].normal[1], planes[k].normal[2], planes[k].d);
    HIPC(ctx, hipMemcpyAsync(ctx->f_models, models, (size_t)nslots * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    // work labels (plane slot / -1 invalid / -2 free): the two halves of f_count (its counts are spent once the slots are assigned) and
    // f_window (only live during the normal-map stage); tile flags: the first bytes of f_dist (normal-map stage only)
    int* w[3] = {reinterpret_cast<int*>(ctx->f_count), reinterpret_cast<int*>(ctx->f_count) + n, ctx->f_window};
    const int tiles_y = (rows + kRefTH - 1) / kRefTH, tiles_x = (cols + 63) / 64;
    unsigned char* tile_free = reinterpret_cast<unsigned char*>(ctx->f_dist);
    HIPC(ctx, hipMemsetAsync(tile_free, 0, (size_t)tiles_x * tiles_y, ctx->stream));
    // X (w[1]) starts as the pass labels (written by the same launch) and is relaxed in place; pass 2 starts from a copy of pass 1's result (w[2])
    hipLaunchKernelGGL(k_f360_refine_init, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->f_label, ctx->f_slot_of_root, ctx->f_models, n, cols, tiles_x,
                       w[0], w[1], tile_free, d_changed);
    const dim3 g(tiles_x, (tiles_y + kRefWaves - 1) / kRefWaves), b(64 * kRefWaves);
    const int n_lds = nslots <= 2048 ? nslots : 0;         // plane models staged in LDS when they fit 32 KB
    const size_t lds_bytes = (size_t)n_lds * sizeof(float4);
    int sweeps = 0;
    for (int pass = 1; pass <= 2; ++pass) {
        const int* W0 = pass == 1 ? w[0] : w[2];
        if (pass == 2) HIPC(ctx, hipMemcpyAsync(w[2], w[1], (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
        bool converged = false;
        // A launch relaxes until the growth chains stop moving (its waves poll their rings, k_f360_refine_tile); the host looks at the
        // "something changed" flag of every second launch (a check costs a stream synchronisation): the run ends with a launch that
        // changed nothing.
        constexpr int kPerCheck = 2;
        static const int kPolls = [] { const char* e = knobs::debug("RGBD360_REFINE_POLLS"); return e ? atoi(e) : 4096; }();      // (512 until round 4: a launch whose growth chains were still moving gave up after 0.34 ms and cost a second round of launches + a host synchronisation)
        static const int kQuiet = [] { const char* e = knobs::debug("RGBD360_REFINE_QUIET"); return e ? atoi(e) : 32; }();
        const int max_rounds = tiles_x + tiles_y + 8;      // a tile is final once its predecessor tiles are: one launch per tile at worst
        for (int round = 0; round < max_rounds && !converged; ++round) {
            volatile int* flags = ctx->f_flags_host;
            for (int k = 0; k < kPerCheck; ++k) flags[k] = 0;
            for (int k = 0; k < kPerCheck; ++k) {
#define REFINE_TILE(P, L) hipLaunchKernelGGL((k_f360_refine_tile<P, L>), g, b, lds_bytes, ctx->stream, ctx->f_xyz, W0, w[1], ctx->f_models, ctx->f_refine_dist, rows, cols, \
                                             tiles_y, tile_free, ctx->f_flags_host + k, d_activity, kPolls, kQuiet, n_lds)
                if (pass == 1) {
                    if (n_lds > 0) REFINE_TILE(1, true); else REFINE_TILE(1, false);
                } else {
                    if (n_lds > 0) REFINE_TILE(2, true); else REFINE_TILE(2, false);
                }
#undef REFINE_TILE
                ++sweeps;
            }
            HIPC(ctx, hipGetLastError());
            HIPC(ctx, hostwait::tag_and_wait(ctx->tag, ctx->stream));      // (a tag kernel + host spin: ~10 us less than hipStreamSynchronize, host_wait.h)
            converged = flags[kPerCheck - 1] == 0;      // a launch that changes nothing is a fixed point: every later one repeats it
        }
        if (!converged) return fail(ctx, -7, "plane refinement did not converge");
        if (knobs::debug("RGBD360_REFINE_DEBUG")) {
            int act = 0;
            hipMemcpy(&act, d_activity, sizeof(int), hipMemcpyDeviceToHost);
            int nfree = 0;
            std::vector<unsigned char> tf((size_t)tiles_x * tiles_y);
            hipMemcpy(tf.data(), tile_free, tf.size(), hipMemcpyDeviceToHost);
            for (unsigned char v : tf) nfree += v ? 1 : 0;
            for (int ty = 0; ty < tiles_y; ++ty) {
                std::string line;
                for (int tx = 0; tx < tiles_x; ++tx) { const int v = tf[(size_t)ty * tiles_x + tx]; line += v == 0 ? '.' : (v == 1 ? 'o' : (v < 11 ? char('0' + v - 1) : '#')); }
                fprintf(stderr, "%s\n", line.c_str());
            }
            fprintf(stderr, "[refine dbg] pass %d: activity (steps that changed a label, cumulative) %d, free tiles %d of %d, sweeps %d\n", pass, act, nfree, tiles_x * tiles_y, sweeps);
        }
    }
    const int* W0 = w[1];
    hipLaunchKernelGGL(k_f360_refine_commit, dim3((n + kCommitThreads * kCommitPerThread - 1) / (kCommitThreads * kCommitPerThread)), dim3(kCommitThreads), 0, ctx->stream, ctx->f_xyz, ctx->f_label, w[0], W0, ctx->f_root_of_slot, n,
                       ctx->f_count_of_slot, ctx->f_mom, kF360MaxSlots, d_changed);
    hipLaunchKernelGGL(k_f360_mom_reduce, dim3((kF360MaxSlots * 9 + 255) / 256), dim3(256), 0, ctx->stream, ctx->f_mom, ctx->f_nslots, kF360MaxSlots,
                       ctx->f_root_of_slot, ctx->f_count_of_slot, ctx->f_pack_host, (const int*)d_changed, ctx->f_flags_host + kFlags);
    // the contour PCL hands to calcConvexHull is that of the REFINED region, projected with the plane `segment` fitted: extremes of the
    // committed labels against the frames k_f360_slot_frames left before the refinement
    if (const int rc_h = launch_hull(ctx, rows, cols, /*clear_first=*/true)) return rc_h;
    launch_colour(ctx, rows, cols);          // the colour of the REFINED inlier sets (Frame360.h:1045-1046 run on the refined regions); a no-op without a colour image
    HIPC(ctx, hipGetLastError());
    HIPC(ctx, hostwait::tag_and_wait(ctx->tag, ctx->stream));      // (a tag kernel + host spin: ~10 us less than hipStreamSynchronize, host_wait.h)
    ctx->f_refine_changed = ctx->f_flags_host[kFlags];
    ctx->f_refine_sweeps = sweeps;
    // count and the extent descriptors of the grown inlier sets (Frame360.h:1010-1037 derives them from the refined inlier cloud);
    // centroid / normal / d / curvature stay those of `segment`, as PCL's PlanarRegion keeps them
    const F360SlotRecord* recs = reinterpret_cast<const F360SlotRecord*>(ctx->f_pack_host + kF360PackHeader);
    for (size_t k = 0; k < planes.size(); ++k) {
        const F360SlotRecord& R = recs[plane_slot[k]];
        double m[9];
        for (int q = 0; q < 9; ++q) m[q] = (double)(long long)R.mom[q] / kMomScale;
        const double N = R.count;
        const double cx = m[0] / N, cy = m[1] / N, cz = m[2] / N;
        const double C[3][3] = {{m[3] / N - cx * cx, m[4] / N - cx * cy, m[5] / N - cx * cz},
                                {m[4] / N - cx * cy, m[6] / N - cy * cy, m[7] / N - cy * cz},
                                {m[5] / N - cx * cz, m[7] / N - cy * cz, m[8] / N - cz * cz}};
        double evs[3], vecs[3][3];
        sorted_eigen3(C, evs, vecs);
        rgbd360_plane& P = planes[k];
        P.count = R.count;
        const double l1 = std::max(evs[1], 0.0), l2 = std::max(evs[2], 0.0);
        P.area_moment = (float)(12.0 * sqrt(l1 * l2));
        P.elongation = (float)(l1 > 0 ? sqrt(l2 / l1) : INFINITY);
        for (int q = 0; q < 3; ++q) P.ppal_dir[q] = (float)vecs[2][q];
        apply_hull(P, hull_records(ctx)[plane_slot[k]]);
        apply_colour(ctx, P, plane_slot[k]);
    }
    return 0;
}

int f360_planes_dev(F360State* ctx, int rows, int cols, int min_inliers, float angular_threshold, float distance_threshold,
                    float max_curvature, int depth_mode, rgbd360_plane* planes, int max_planes, int* n_planes) {
    using namespace f360;
    const int n = rows * cols;
    uint8_t* flags = ctx->f_change;
    hipLaunchKernelGGL(k_f360_link_flags, dim3((cols + kLinkTW - 1) / kLinkTW, (rows + kLinkTH - 1) / kLinkTH), dim3(kLinkTW), 0, ctx->stream,
                       ctx->f_xyz, ctx->f_normals, rows, cols, cosf(angular_threshold), distance_threshold, depth_mode, flags);
    // run starts as compact per-row lists for the root pass and the slot assignment: in f_window (the normal-map stage's window plane,
    // spent by now; the refinement takes it over later) and, for the counts, in the tail of f_hd (the depth-change mask is spent once the distance map exists)
    int* run_starts = ctx->f_window;        // (no other use of this plane since round 6; f_slot_of_root stays free for k_f360_assign_list's writes)
    int* n_run_starts = reinterpret_cast<int*>(ctx->f_hd + (((size_t)n + 15) & ~(size_t)15));      // n + 4 rows <= 3 n + 64 bytes
    const bool seg_rows = cols % 4 == 0 && cols >= 1024 && cols <= kRunSegs * kRunSegSteps * 256;       // (flags and labels are hipMalloc'ed: rows of whole, aligned dwords)
    if (seg_rows)
        hipLaunchKernelGGL(k_f360_ccl_runs_seg, dim3(rows), dim3(64 * kRunSegs), 0, ctx->stream, flags, rows, cols, ctx->f_label, run_starts, n_run_starts);
    else
        hipLaunchKernelGGL(k_f360_ccl_runs, dim3((rows + kRunRowsPerBlock - 1) / kRunRowsPerBlock), dim3(64 * kRunRowsPerBlock), 0, ctx->stream,
                           flags, rows, cols, ctx->f_label, run_starts, n_run_starts);
    {
        const dim3 gb((cols + kBandCols - 1) / kBandCols, (rows + kBandRows - 1) / kBandRows);
        if (kBandGroupsMax >= 16 && gb.x * gb.y >= 1536)
            hipLaunchKernelGGL((k_f360_ccl_merge_band<kBandGroupsMax / 2>), gb, dim3(kBandCols * (kBandGroupsMax / 2)), 0, ctx->stream, flags, rows, cols, ctx->f_label);
        else
            hipLaunchKernelGGL((k_f360_ccl_merge_band<kBandGroupsMax>), gb, dim3(kBandCols * kBandGroupsMax), 0, ctx->stream, flags, rows, cols, ctx->f_label);
    }
    constexpr int kTopLevel = kBandRows == 64 ? kBandLevels : kBandLevels + 2;      // 64-row bands: every 64th row in one launch behind them
    for (int level = kBandLevels; level <= kTopLevel && (1 << level) < rows; ++level) {
        const bool all_above = level == kTopLevel;
        const int n_rows = all_above ? (rows - 1) / (1 << level) : (rows - 1 - (1 << level)) / (2 << level) + 1;
        hipLaunchKernelGGL(k_f360_ccl_merge_level, dim3((cols + 255) / 256, n_rows), dim3(256), 0, ctx->stream, flags, rows, cols,
                           ctx->f_label, level, all_above ? 1 : 0);
    }
#ifdef F360_DEBUG_COUNTERS
    {
        unsigned long long h[8];
        hipStreamSynchronize(ctx->stream);
        hipMemcpyFromSymbol(h, HIP_SYMBOL(f360::g_dbg), sizeof(h));
        fprintf(stderr, "[f360 dbg] after merge: unions %llu find-hops %llu atomicMin %llu longest walk %llu\n", h[0], h[1], h[2], h[3]);
    }
#endif
    hipLaunchKernelGGL(k_f360_ccl_roots_list, dim3(rows), dim3(kRootsThreads), 0, ctx->stream, run_starts, n_run_starts, cols, ctx->f_label, ctx->f_count);
#ifdef F360_DEBUG_COUNTERS
    {
        unsigned long long h[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        hipStreamSynchronize(ctx->stream);
        hipMemcpyFromSymbol(h, HIP_SYMBOL(f360::g_dbg), sizeof(h));
        fprintf(stderr, "[f360 dbg] after roots: find-hops %llu (cumulative) longest walk %llu\n", h[1], h[3]);
        hipMemcpyToSymbol(HIP_SYMBOL(f360::g_dbg), z, sizeof(z));
    }
#endif
    hipLaunchKernelGGL(k_f360_finish_count, dim3((n + kCntThreads * kCntPerThread - 1) / (kCntThreads * kCntPerThread)), dim3(kCntThreads), 0,
                       ctx->stream, flags, n, ctx->f_label, ctx->f_count, ctx->f_nslots);
    const dim3 bagg(kAggThreads);
    hipLaunchKernelGGL(k_f360_assign_list, dim3(rows), dim3(kAssignThreads), 0, ctx->stream, run_starts, n_run_starts, cols, ctx->f_label, ctx->f_count,
                       min_inliers, kF360MaxSlots, ctx->f_slot_of_root, ctx->f_root_of_slot, ctx->f_count_of_slot, ctx->f_nslots, ctx->f_mom,
                       f360::kMomReplicas);
    const dim3 gmom((n + kAggThreads * kMomPerThread - 1) / (kAggThreads * kMomPerThread));
    hipLaunchKernelGGL(k_f360_moments, gmom, bagg, 0, ctx->stream, ctx->f_xyz, ctx->f_label, ctx->f_slot_of_root, n, ctx->f_mom, kF360MaxSlots);
    // hull stage: per slot the in-plane frame (and an empty extremes row); the extremes themselves now, or -- with the refinement
    // switched on -- once the refined labels are committed (f360_refine_dev)
    // The header + records go straight into pinned host memory (a few KB over PCIe): no copy to enqueue, one wait.  (Four copies into
    // pageable vectors with two waits used to cost ~0.1 ms of the 0.45 ms call.)  The frame kernel packs them on its way (it sums the
    // moment replicas anyway); k_f360_mom_reduce only runs behind the refinement's commit, which changes the sums.
    hipLaunchKernelGGL(k_f360_slot_frames, dim3(256), dim3(64), 0, ctx->stream, ctx->f_mom, ctx->f_nslots, kF360MaxSlots,
                       ctx->f_count_of_slot, ctx->f_frames, ctx->f_ext, ctx->f_root_of_slot, ctx->f_pack_host);
    ctx->f_col_ran = false;
    if (!ctx->f_refine) {
        if (const int rc_h = launch_hull(ctx, rows, cols, /*clear_first=*/false)) return rc_h;
        launch_colour(ctx, rows, cols);       // (a no-op without a colour image of this geometry)
    }
    if (ctx->f_stage_timing) {
        hipEventRecord(ctx->f_stage_ev[3], ctx->stream);
        ctx->f_stage_valid = true;
    }
    HIPC(ctx, hipGetLastError());
    // one tag kernel behind the chain + a host spin (~10 us less than hipStreamSynchronize, host_wait.h).  Until round 5 the chain's last
    // kernel published the tag from its last block: 256 blocks x a system-scope fence cost 10-22 us more than this launch.
    HIPC(ctx, hostwait::tag_and_wait(ctx->tag, ctx->stream));
    const int nslots = *reinterpret_cast<const volatile int*>(ctx->f_pack_host);
    if (nslots > kF360MaxSlots) return fail(ctx, -7, "more than 4096 regions exceed min_inliers");
    const F360SlotRecord* recs = reinterpret_cast<const F360SlotRecord*>(ctx->f_pack_host + kF360PackHeader);
    std::vector<int> roots(nslots), counts(nslots);
    std::vector<double> mom((size_t)nslots * 9);
    for (int s = 0; s < nslots; ++s) {
        roots[s] = recs[s].root;
        counts[s] = recs[s].count;
        for (int k = 0; k < 9; ++k) mom[(size_t)s * 9 + k] = (double)(long long)recs[s].mom[k] / kMomScale;     // fixed point -> metres
    }
    std::vector<int> order(nslots);
    for (int s = 0; s < nslots; ++s) order[s] = s;
    std::sort(order.begin(), order.end(), [&](int a, int c) { return roots[a] < roots[c]; });   // PCL's order: by first pixel
    // Planes are built for every region first: when more than max_planes pass the curvature filter the LARGEST ones are kept
    // (still in PCL's order) instead of the first -- the regions lowest in the image (typically the floor) used to be the ones
    // cut -- and the total is remembered for rgbd360_planes_available so that an adapter can grow its buffer and call again.
    std::vector<rgbd360_plane> all;
    std::vector<int> all_slot;      // region slot of every plane (its sums live in row `slot` of the moment table)
    all.reserve(nslots);
    for (int oi = 0; oi < nslots; ++oi) {
        const int s = order[oi];
        const double* m = &mom[(size_t)s * 9];
        const double N = counts[s];
        const double cx = m[0] / N, cy = m[1] / N, cz = m[2] / N;
        const double C[3][3] = {{m[3] / N - cx * cx, m[4] / N - cx * cy, m[5] / N - cx * cz},
                                {m[4] / N - cx * cy, m[6] / N - cy * cy, m[7] / N - cy * cz},
                                {m[5] / N - cx * cz, m[7] / N - cy * cz, m[8] / N - cz * cz}};
        // the fixed-point sums wrap beyond N r^2 = 2^63 / kMomScale (frame360_kernels.h): a wrapped sum of squares is off by
        // 6.9e10 / N m^2, i.e. negative or absurd -- refused, not returned
        if (!(m[3] >= 0.0 && m[6] >= 0.0 && m[8] >= 0.0) || C[0][0] < -1e-3 || C[1][1] < -1e-3 || C[2][2] < -1e-3)
            return fail(ctx, -8, "region moments out of range (points beyond ~60 m over a whole frame)");
        double evs[3], vecs[3][3];
        sorted_eigen3(C, evs, vecs);
        const double ev = evs[0];
        double* v = vecs[0];
        double d = -(v[0] * cx + v[1] * cy + v[2] * cz);
        if ((-cx) * v[0] + (-cy) * v[1] + (-cz) * v[2] < 0) {     // orient towards the viewpoint (origin)
            v[0] = -v[0]; v[1] = -v[1]; v[2] = -v[2];
            d = -(v[0] * cx + v[1] * cy + v[2] * cz);
        }
        const double tr = C[0][0] + C[1][1] + C[2][2];
        const double curvature = tr != 0 ? fabs(ev / tr) : 0;
        if (!(curvature < max_curvature)) continue;
        all.emplace_back();
        all_slot.push_back(s);
        rgbd360_plane& P = all.back();
        P.centroid[0] = (float)cx; P.centroid[1] = (float)cy; P.centroid[2] = (float)cz;
        P.normal[0] = (float)v[0]; P.normal[1] = (float)v[1]; P.normal[2] = (float)v[2];
        P.d = (float)d;
        P.curvature = (float)curvature;
        P.count = counts[s];
        P.root = roots[s];
        const double l1 = std::max(evs[1], 0.0), l2 = std::max(evs[2], 0.0);    // in-plane moments (rgbd360_hip.h)
        P.area_moment = (float)(12.0 * sqrt(l1 * l2));
        P.elongation = (float)(l1 > 0 ? sqrt(l2 / l1) : INFINITY);
        for (int k = 0; k < 3; ++k) P.ppal_dir[k] = (float)vecs[2][k];
        apply_colour(ctx, P, s);      // (with the refinement on: empty here, filled by f360_refine_dev below)
        if (!ctx->f_refine) apply_hull(P, hull_records(ctx)[s]);
        else {                   // filled by f360_refine_dev below
            P.area = P.area_moment;
            P.hull_points = 0;
            P.hull_n = 0;
            for (int k = 0; k < 3; ++k) P.center_hull[k] = P.centroid[k];
        }
    }
    ctx->f_refine_changed = ctx->f_refine_sweeps = 0;
    if (ctx->f_refine && !all.empty()) {
        const int rc = f360_refine_dev(ctx, rows, cols, nslots, all, all_slot);
        if (rc) return rc;
    }
    ctx->f_planes_available = (int)all.size();
    int np = 0;
    if ((int)all.size() <= max_planes) {
        for (const rgbd360_plane& P : all) planes[np++] = P;
    } else {
        std::vector<int> by_count(all.size());
        for (size_t k = 0; k < all.size(); ++k) by_count[k] = (int)k;
        std::stable_sort(by_count.begin(), by_count.end(), [&](int a, int c) { return all[a].count > all[c].count; });
        std::vector<char> keep(all.size(), 0);
        for (int k = 0; k < max_planes; ++k) keep[by_count[k]] = 1;
        for (size_t k = 0; k < all.size(); ++k)
            if (keep[k]) planes[np++] = all[k];
    }
    *n_planes = np;
    return 0;
}
// organised cloud of one spherical depth image -> ctx->f_xyz (device); the per-row/column sin/cos tables follow the
// reference's float expressions and are computed on the host (rows + cols values)
// defer_to_edge_kernel: tables and depth are put in place, the points themselves are left to k_f360_edge_bits<true> (the next stage
// of rgbd360_frame_planes), which forms them anyway
int sphere_cloud_dev(F360State* ctx, const void* depth, size_t depth_step, int depth_type, int rows, int cols, int convention,
                     bool depth_on_device = false, bool defer_to_edge_kernel = false) {
    if (convention < 0 || convention > 2 || (depth_type != 0 && depth_type != 1)) return fail(ctx, -1, "bad arguments");
    const bool tables_resident = ctx->f_tab && ctx->f_tab_rows == rows && ctx->f_tab_cols == cols && ctx->f_tab_conv == convention;
    std::vector<float> st(cols), ct(cols), sp(rows), cp(rows);
    if (tables_resident) {          // the angle tables of this geometry are already on the device
    } else if (convention == 0) {   // Frame360.h:562-585
        const float angle_pixel(cols / (2 * kPI));
        const float angle_pixel_inv(1 / angle_pixel);
        const float offset_phi = kPI * 31.5 / 180;
        for (int r = 0; r < rows; ++r) {
            float phi_i = offset_phi - r * angle_pixel_inv;
            sp[r] = sinf(phi_i);
            cp[r] = cosf(phi_i);
        }
        for (int c = 0; c < cols; ++c) {
            float theta_i = c * angle_pixel_inv;
            st[c] = sinf(theta_i);
            ct[c] = cosf(theta_i);
        }
    } else if (convention == 1) {   // Frame360_stereo.h:470-490
        const float step_theta = 2 * kPI / cols;
        const float step_phi = step_theta;
        const int start_phi = 166;
        for (int r = 0; r < rows; ++r) {
            float phi = (r + start_phi) * step_phi - kPI / 2;
            cp[r] = cosf(phi);
            sp[r] = sinf(phi);
        }
        for (int c = 0; c < cols; ++c) {
            float theta = c * step_theta - kPI;
            st[c] = sinf(theta);
            ct[c] = cosf(theta);
        }
    } else {                        // RPI.h:4556-4571
        const float angle_res = 2 * kPI / cols;
        const float half_nRows = 0.5 * rows - 0.5;
        for (int c = 0; c < cols; ++c) {
            float theta = c * angle_res;
            st[c] = sinf(theta);
            ct[c] = cosf(theta);
        }
        for (int r = 0; r < rows; ++r) {
            float phi = (half_nRows - r) * angle_res;
            sp[r] = sinf(phi);
            cp[r] = cosf(phi);
        }
    }
    const size_t dpx = depth_type == 0 ? 2 : 4;
    if (!tables_resident) {
        const size_t ntab = (size_t)2 * cols + 2 * rows;
        if (ctx->f_tab_n < ntab) {
            hipFree(ctx->f_tab);
            ctx->f_tab = nullptr;
            ctx->f_tab_n = 0;
            HIPC(ctx, hipMalloc(&ctx->f_tab, ntab * sizeof(float)));
            ctx->f_tab_n = ntab;
        }
        std::vector<float> tab;
        tab.insert(tab.end(), st.begin(), st.end());
        tab.insert(tab.end(), ct.begin(), ct.end());
        tab.insert(tab.end(), sp.begin(), sp.end());
        tab.insert(tab.end(), cp.begin(), cp.end());
        ctx->f_tab_conv = -1;
        HIPC(ctx, hipMemcpyAsync(ctx->f_tab, tab.data(), ntab * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        HIPC(ctx, hipStreamSynchronize(ctx->stream));      // `tab` (pageable) must outlive the copy
        ctx->f_tab_rows = rows; ctx->f_tab_cols = cols; ctx->f_tab_conv = convention;
    }
    const void* d_depth = depth;
    size_t d_step = depth_step;
    if (!depth_on_device) {
        HIPC(ctx, hipMemcpy2DAsync(ctx->f_depth_raw, (size_t)cols * dpx, depth, depth_step, (size_t)cols * dpx, rows, hipMemcpyHostToDevice,
                                   ctx->stream));
        d_depth = ctx->f_depth_raw;
        d_step = (size_t)cols * dpx;
    }
    float* d_tab = ctx->f_tab;
    ctx->f_cloud_pending.depth = nullptr;
    if (defer_to_edge_kernel) {
        ctx->f_cloud_pending = {d_depth, d_step, depth_type, convention, d_tab, d_tab + cols, d_tab + 2 * cols, d_tab + 2 * cols + rows};
        if (!depth_on_device) HIPC(ctx, hipStreamSynchronize(ctx->stream));      // the caller may reuse its host image
        return 0;
    }
    // consecutive lanes = consecutive pixels, one 12-byte store per lane, four pixels (256 apart) per thread
    hipLaunchKernelGGL(k_sphere_cloud_s4, dim3((cols + 1023) / 1024, rows), dim3(256), 0, ctx->stream, d_depth, d_step, depth_type, rows, cols,
                       convention, d_tab, d_tab + cols, d_tab + 2 * cols, d_tab + 2 * cols + rows, ctx->f_xyz);
    HIPC(ctx, hipGetLastError());
    if (!depth_on_device) HIPC(ctx, hipStreamSynchronize(ctx->stream));      // the caller may reuse its host image
    return 0;
}
}  // namespace

extern "C" int rgbd360_sphere_cloud(rgbd360_ctx* ctx_, const void* depth, size_t depth_step, int depth_type, int rows, int cols,
                         int convention, float* host_out_xyz) {
    F360_ENTER(ctx_);
    if (!ctx || !depth || !host_out_xyz) return -1;
    if (rows < 1 || cols < 1 || (long long)rows * cols >= (1ll << 30)) return fail(ctx, -1, "bad image size");
    hipSetDevice(ctx->p.device);
    const size_t n = (size_t)rows * cols;
    int rc = f360_ensure(ctx, n);
    if (rc) return rc;
    rc = sphere_cloud_dev(ctx, depth, depth_step, depth_type, rows, cols, convention);
    if (rc) return rc;
    HIPC(ctx, hipMemcpyAsync(host_out_xyz, ctx->f_xyz, n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int rgbd360_normals(rgbd360_ctx* ctx_, const float* xyz, int rows, int cols, float max_depth_change_factor,
                               float normal_smoothing_size, int depth_mode, float* normals_out) {
    F360_ENTER(ctx_);
    if (!ctx || !xyz || !normals_out) return -1;
    if (rows < 3 || cols < 3 || (long long)rows * cols >= (1ll << 30)) return fail(ctx, -1, "bad image size");
    hipSetDevice(ctx->p.device);
    const size_t n = (size_t)rows * cols;
    int rc = f360_ensure(ctx, n);
    if (rc) return rc;
    HIPC(ctx, hipMemcpyAsync(ctx->f_xyz, xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    rc = f360_normals_dev(ctx, rows, cols, max_depth_change_factor, normal_smoothing_size, depth_mode);
    if (rc) return rc;
    HIPC(ctx, hipMemcpyAsync(normals_out, ctx->f_normals, n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int rgbd360_bilateral_filter(rgbd360_ctx* ctx_, const float* xyz, int rows, int cols, float sigma_s, float sigma_r,
                                       float* xyz_out) {
    F360_ENTER(ctx_);
    if (!ctx || !xyz || !xyz_out) return -1;
    if (rows < 1 || cols < 1 || (long long)rows * cols >= (1ll << 30)) return fail(ctx, -1, "bad image size");
    hipSetDevice(ctx->p.device);
    const size_t n = (size_t)rows * cols;
    int rc = f360_ensure(ctx, n);
    if (rc) return rc;
    HIPC(ctx, hipMemcpyAsync(ctx->f_xyz, xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    rc = f360_bilateral_dev(ctx, rows, cols, sigma_s, sigma_r);
    if (rc) return rc;
    HIPC(ctx, hipMemcpyAsync(xyz_out, ctx->f_xyz, n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int rgbd360_distance_map(rgbd360_ctx* ctx_, const float* xyz, int rows, int cols, float max_depth_change_factor,
                                    int depth_mode, float* dist_out) {
    F360_ENTER(ctx_);
    if (!ctx || !xyz || !dist_out) return -1;
    if (rows < 3 || cols < 3 || (long long)rows * cols >= (1ll << 30)) return fail(ctx, -1, "bad image size");
    hipSetDevice(ctx->p.device);
    const size_t n = (size_t)rows * cols;
    int rc = f360_ensure(ctx, n);
    if (rc) return rc;
    HIPC(ctx, hipMemcpyAsync(ctx->f_xyz, xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    launch_distance_map(ctx, rows, cols, max_depth_change_factor, depth_mode);
    HIPC(ctx, hipGetLastError());
    HIPC(ctx, hipMemcpyAsync(dist_out, ctx->f_dist, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// measurement: HIP events at the stage boundaries of the context's later frame_planes calls (rgbd360_hip_diag.h)
extern "C" int rgbd360_frame_planes_stage_timing(rgbd360_ctx* ctx_, int on) {
    F360_ENTER(ctx_);
    if (!ctx) return -1;
    hipSetDevice(ctx->p.device);
    if (on)
        for (hipEvent_t& e : ctx->f_stage_ev)
            if (!e) HIPC(ctx, hipEventCreate(&e));
    ctx->f_stage_timing = on != 0;
    ctx->f_stage_valid = false;
    return 0;
}
extern "C" int rgbd360_frame_planes_stage_times(rgbd360_ctx* ctx_, float us[3]) {
    F360_ENTER(ctx_);
    if (!ctx || !us) return -1;
    if (!ctx->f_stage_timing || !ctx->f_stage_valid) return fail(ctx, -1, "no timed frame_planes call (rgbd360_frame_planes_stage_timing(ctx, 1) first; the refinement must be off)");
    hipSetDevice(ctx->p.device);
    HIPC(ctx, hipEventSynchronize(ctx->f_stage_ev[3]));
    for (int k = 0; k < 3; ++k) {
        float ms = 0.f;
        HIPC(ctx, hipEventElapsedTime(&ms, ctx->f_stage_ev[k], ctx->f_stage_ev[k + 1]));
        us[k] = ms * 1000.f;
    }
    return 0;
}

extern "C" int rgbd360_planes_available(rgbd360_ctx* ctx_) {
    F360_ENTER(ctx_);
    return ctx->f_planes_available;
}

extern "C" int rgbd360_set_plane_refinement(rgbd360_ctx* ctx_, int enabled, float distance_threshold) {
    F360_ENTER(ctx_);
    if (!ctx) return -1;
    if (enabled && !(distance_threshold > 0.f)) return fail(ctx, -1, "the refinement distance threshold must be positive");
    ctx->f_refine = enabled ? 1 : 0;
    if (enabled) ctx->f_refine_dist = distance_threshold;
    return 0;
}
extern "C" int rgbd360_set_plane_color_image(rgbd360_ctx* ctx_, const uint8_t* rgb, size_t rgb_step, int rows, int cols, int step, int on_device) {
    F360_ENTER(ctx_);
    if (!ctx) return -1;
    hipSetDevice(ctx->p.device);
    if (!rgb) {
        ctx->f_col_img = {nullptr, 0, 1};
        ctx->f_col_rows = ctx->f_col_cols = 0;
        return 0;
    }
    if (rows < 1 || cols < 1 || step < 1 || step > 16 || rgb_step < (size_t)cols * 3) return fail(ctx, -1, "bad colour image geometry");
    if (on_device) {
        ctx->f_col_img = {rgb, rgb_step, step};
    } else {
        const size_t bytes = (size_t)rows * cols * 3;
        HIPC(ctx, hipStreamSynchronize(ctx->stream));       // a plane call still reading the previous copy
        if (ctx->f_col_owned_bytes < bytes) {
            hipFree(ctx->f_col_owned);
            ctx->f_col_owned = nullptr;
            ctx->f_col_owned_bytes = 0;
            HIPC(ctx, hipMalloc(&ctx->f_col_owned, bytes));
            ctx->f_col_owned_bytes = bytes;
        }
        HIPC(ctx, hipMemcpy2D(ctx->f_col_owned, (size_t)cols * 3, rgb, rgb_step, (size_t)cols * 3, rows, hipMemcpyHostToDevice));
        ctx->f_col_img = {ctx->f_col_owned, (size_t)cols * 3, step};
    }
    ctx->f_col_rows = rows; ctx->f_col_cols = cols;
    return 0;
}
extern "C" int rgbd360_plane_refinement_stats(rgbd360_ctx* ctx_, int* pixels_relabelled, int* sweeps) {
    F360_ENTER(ctx_);
    if (!ctx) return -1;
    if (pixels_relabelled) *pixels_relabelled = ctx->f_refine_changed;
    if (sweeps) *sweeps = ctx->f_refine_sweeps;
    return 0;
}

extern "C" int rgbd360_plane_fit(rgbd360_ctx* ctx_, const float* xyz, const float* normals, int rows, int cols, int min_inliers,
                                 float angular_threshold, float distance_threshold, float max_curvature, int depth_mode,
                                 int32_t* labels_out, rgbd360_plane* planes_out, int max_planes, int* n_planes_out) {
    F360_ENTER(ctx_);
    if (!ctx || !xyz || !normals || !planes_out || !n_planes_out || max_planes < 1) return -1;
    if (rows < 2 || cols < 2 || (long long)rows * cols >= (1ll << 30)) return fail(ctx, -1, "bad image size");
    hipSetDevice(ctx->p.device);
    const size_t n = (size_t)rows * cols;
    int rc = f360_ensure(ctx, n);
    if (rc) return rc;
    HIPC(ctx, hipMemcpyAsync(ctx->f_xyz, xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    HIPC(ctx, hipMemcpyAsync(ctx->f_normals, normals, n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    rc = f360_planes_dev(ctx, rows, cols, min_inliers, angular_threshold, distance_threshold, max_curvature, depth_mode, planes_out,
                         max_planes, n_planes_out);
    if (rc) return rc;
    if (labels_out) HIPC(ctx, hipMemcpy(labels_out, ctx->f_label, n * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

static int frame_planes_impl(F360State* ctx, const void* depth, size_t depth_step, int depth_type, int rows, int cols,
                             int convention, float max_depth_change_factor, float normal_smoothing_size, int min_inliers,
                             float angular_threshold, float distance_threshold, float max_curvature, int depth_mode,
                             float* xyz_out, float* normals_out, int32_t* labels_out, rgbd360_plane* planes_out,
                             int max_planes, int* n_planes_out, bool depth_on_device) {
    if (!ctx || !depth || !planes_out || !n_planes_out || max_planes < 1) return -1;
    if (rows < 3 || cols < 3 || (long long)rows * cols >= (1ll << 30)) return fail(ctx, -1, "bad image size");
    hipSetDevice(ctx->p.device);
    const size_t n = (size_t)rows * cols;
    int rc = f360_ensure(ctx, n);
    if (rc) return rc;
    // the cloud stays on the device; a host copy is only made when asked for
    ctx->f_stage_valid = false;
    if (ctx->f_stage_timing) hipEventRecord(ctx->f_stage_ev[0], ctx->stream);
    rc = sphere_cloud_dev(ctx, depth, depth_step, depth_type, rows, cols, convention, depth_on_device, /*defer_to_edge_kernel=*/true);
    if (rc) return rc;
    rc = f360_normals_dev(ctx, rows, cols, max_depth_change_factor, normal_smoothing_size, depth_mode);
    if (rc) {
        ctx->f_cloud_pending.depth = nullptr;
        return rc;
    }
    if (xyz_out) HIPC(ctx, hipMemcpyAsync(xyz_out, ctx->f_xyz, n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    rc = f360_planes_dev(ctx, rows, cols, min_inliers, angular_threshold, distance_threshold, max_curvature, depth_mode, planes_out,
                         max_planes, n_planes_out);
    if (rc) return rc;
    if (normals_out) HIPC(ctx, hipMemcpy(normals_out, ctx->f_normals, n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (labels_out) HIPC(ctx, hipMemcpy(labels_out, ctx->f_label, n * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

// one sensor's cloud (pinhole + median down-sampling) from a host depth image into ctx->f_xyz (device)
static int sensor_cloud_upload(F360State* ctx, const void* depth, size_t depth_step, int depth_type, int rows, int cols, int step, float min_depth,
                               float max_depth) {
    const size_t dpx = depth_type == 0 ? 2 : 4;
    if (rows < 1 || cols < 1 || step < 1 || step > 4 || rows / step < 1 || cols / step < 1 || depth_step < (size_t)cols * dpx ||
        (depth_type != 0 && depth_type != 1) || (long long)rows * cols >= (1ll << 30))
        return fail(ctx, -1, "bad arguments");
    hipSetDevice(ctx->p.device);
    const size_t n = (size_t)rows * cols;
    int rc = f360_ensure(ctx, n);
    if (rc) return rc;
    HIPC(ctx, hipMemcpy2DAsync(ctx->f_depth_raw, (size_t)cols * dpx, depth, depth_step, (size_t)cols * dpx, rows, hipMemcpyHostToDevice, ctx->stream));
    f360::SensorCloudArgs a;
    a.rows = rows; a.cols = cols; a.step = step;
    a.depth_f32 = depth_type;
    const float res_factor_VGA = cols / 640.0;                       // CloudRGBD.h:118-123
    const float focal_length = 525 * res_factor_VGA;
    a.inv_fx = 1.f / focal_length; a.inv_fy = 1.f / focal_length;
    a.ox = cols / 2 - 0.5; a.oy = rows / 2 - 0.5;
    a.min_depth = min_depth; a.max_depth = max_depth;
    const int on = (rows / step) * (cols / step);
    hipLaunchKernelGGL(f360::k_sensor_cloud, dim3((on + 255) / 256), dim3(256), 0, ctx->stream, ctx->f_depth_raw, (size_t)cols * dpx, a, ctx->f_xyz);
    HIPC(ctx, hipGetLastError());
    return 0;
}

extern "C" int rgbd360_sensor_cloud_ex(rgbd360_ctx* ctx_, const void* depth, size_t depth_step, int depth_type, int rows, int cols, int step,
                                      float min_depth, float max_depth, float* xyz_out) {
    F360_ENTER(ctx_);
    if (!ctx || !depth || !xyz_out) return -1;
    const int rc = sensor_cloud_upload(ctx, depth, depth_step, depth_type, rows, cols, step, min_depth, max_depth);
    if (rc) return rc;
    const size_t on = (size_t)(rows / step) * (cols / step);
    HIPC(ctx, hipMemcpyAsync(xyz_out, ctx->f_xyz, on * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
extern "C" int rgbd360_sensor_cloud(rgbd360_ctx* ctx_, const uint16_t* depth, size_t depth_step, int rows, int cols, int step, float min_depth,
                                   float max_depth, float* xyz_out) {
    return rgbd360_sensor_cloud_ex(ctx_, depth, depth_step, 0, rows, cols, step, min_depth, max_depth, xyz_out);
}

// the cloud in ctx->f_xyz -> (filter) -> normal map -> regions -> planes, moved by Rt
static int cloud_planes_tail(F360State* ctx, int rows, int cols, float sigma_s, float sigma_r, float max_depth_change_factor,
                             float normal_smoothing_size, int min_inliers, float angular_threshold, float distance_threshold,
                             float max_curvature, int depth_mode, const float Rt[16], rgbd360_plane* planes_out, int max_planes,
                             int* n_planes_out) {
    int rc = 0;
    if (sigma_s > 0.f) {                                             // Frame360.h:493-499
        rc = f360_bilateral_dev(ctx, rows, cols, sigma_s, sigma_r);
        if (rc) return rc;
    }
    rc = f360_normals_dev(ctx, rows, cols, max_depth_change_factor, normal_smoothing_size, depth_mode);      // Frame360.h:949-957
    if (rc) return rc;
    rc = f360_planes_dev(ctx, rows, cols, min_inliers, angular_threshold, distance_threshold, max_curvature, depth_mode, planes_out,
                         max_planes, n_planes_out);                                                         // Frame360.h:958-996
    if (rc) return rc;
    if (Rt) {                                                        // plane.transform(Rt), Frame360.h:1046: sensor -> rig frame
        for (int k = 0; k < *n_planes_out; ++k) {
            rgbd360_plane& P = planes_out[k];
            double nn[3], cc[3], pp[3], hh[3];
            for (int i = 0; i < 3; ++i) {
                nn[i] = cc[i] = pp[i] = hh[i] = 0;
                for (int j = 0; j < 3; ++j) {
                    nn[i] += (double)Rt[j * 4 + i] * P.normal[j];
                    cc[i] += (double)Rt[j * 4 + i] * P.centroid[j];
                    pp[i] += (double)Rt[j * 4 + i] * P.ppal_dir[j];
                    hh[i] += (double)Rt[j * 4 + i] * P.center_hull[j];
                }
                cc[i] += (double)Rt[12 + i];
                hh[i] += (double)Rt[12 + i];
            }
            double dd = -(nn[0] * cc[0] + nn[1] * cc[1] + nn[2] * cc[2]);
            bool flipped = false;
            if (dd < 0) {                                            // keep the normal towards the new origin (Frame360.h:989-993)
                for (int i = 0; i < 3; ++i) nn[i] = -nn[i];
                dd = -dd;
                flipped = true;
            }
            for (int i = 0; i < 3; ++i) {
                P.normal[i] = (float)nn[i];
                P.centroid[i] = (float)cc[i];
                P.ppal_dir[i] = (float)pp[i];
                P.center_hull[i] = (float)hh[i];
            }
            P.d = (float)dd;
            const int hn = std::min(P.hull_n, (int)RGBD360_HULL_MAX);
            for (int v = 0; v < hn; ++v) {                           // the polygon travels with the plane
                double q[3];
                for (int i = 0; i < 3; ++i) q[i] = (double)Rt[0 * 4 + i] * P.hull[v][0] + (double)Rt[1 * 4 + i] * P.hull[v][1] + (double)Rt[2 * 4 + i] * P.hull[v][2] + (double)Rt[12 + i];
                for (int i = 0; i < 3; ++i) P.hull[v][i] = (float)q[i];
            }
            // a rigid motion keeps the polygon's sense about the MOVED normal; where the normal was turned round (the plane lies between the
            // sensor's origin and the rig's) the header's promise -- counter-clockwise seen from the side the normal points to -- needs the
            // vertex order reversed
            if (flipped)
                for (int a = 0, b = hn - 1; a < b; ++a, --b)
                    for (int i = 0; i < 3; ++i) std::swap(P.hull[a][i], P.hull[b][i]);
        }
    }
    return 0;
}

extern "C" int rgbd360_sensor_planes_ex(rgbd360_ctx* ctx_, const void* depth, size_t depth_step, int depth_type, int rows, int cols, int step,
                                       float min_depth, float max_depth, float sigma_s, float sigma_r, float max_depth_change_factor,
                                       float normal_smoothing_size, int min_inliers, float angular_threshold, float distance_threshold,
                                       float max_curvature, const float Rt[16], rgbd360_plane* planes_out, int max_planes, int* n_planes_out) {
    F360_ENTER(ctx_);
    if (!ctx || !depth || !planes_out || !n_planes_out || max_planes < 1) return -1;
    if (step < 1 || rows / step < 3 || cols / step < 3) return fail(ctx, -1, "bad image size");
    const int rc = sensor_cloud_upload(ctx, depth, depth_step, depth_type, rows, cols, step, min_depth, max_depth);
    if (rc) return rc;
    return cloud_planes_tail(ctx, rows / step, cols / step, sigma_s, sigma_r, max_depth_change_factor, normal_smoothing_size, min_inliers,
                             angular_threshold, distance_threshold, max_curvature, /*depth_mode=*/0, Rt, planes_out, max_planes, n_planes_out);
}
extern "C" int rgbd360_sensor_planes(rgbd360_ctx* ctx_, const uint16_t* depth, size_t depth_step, int rows, int cols, int step, float min_depth,
                                    float max_depth, float sigma_s, float sigma_r, float max_depth_change_factor,
                                    float normal_smoothing_size, int min_inliers, float angular_threshold, float distance_threshold,
                                    float max_curvature, const float Rt[16], rgbd360_plane* planes_out, int max_planes, int* n_planes_out) {
    return rgbd360_sensor_planes_ex(ctx_, depth, depth_step, 0, rows, cols, step, min_depth, max_depth, sigma_s, sigma_r, max_depth_change_factor,
                                    normal_smoothing_size, min_inliers, angular_threshold, distance_threshold, max_curvature, Rt, planes_out, max_planes,
                                    n_planes_out);
}

extern "C" int rgbd360_cloud_planes(rgbd360_ctx* ctx_, const float* xyz, int rows, int cols, float sigma_s, float sigma_r,
                                   float max_depth_change_factor, float normal_smoothing_size, int min_inliers, float angular_threshold,
                                   float distance_threshold, float max_curvature, int depth_mode, const float Rt[16],
                                   rgbd360_plane* planes_out, int max_planes, int* n_planes_out) {
    F360_ENTER(ctx_);
    if (!ctx || !xyz || !planes_out || !n_planes_out || max_planes < 1) return -1;
    if (rows < 3 || cols < 3 || (long long)rows * cols >= (1ll << 30)) return fail(ctx, -1, "bad image size");
    hipSetDevice(ctx->p.device);
    const size_t n = (size_t)rows * cols;
    const int rc = f360_ensure(ctx, n);
    if (rc) return rc;
    HIPC(ctx, hipMemcpyAsync(ctx->f_xyz, xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    return cloud_planes_tail(ctx, rows, cols, sigma_s, sigma_r, max_depth_change_factor, normal_smoothing_size, min_inliers, angular_threshold,
                             distance_threshold, max_curvature, depth_mode, Rt, planes_out, max_planes, n_planes_out);
}

extern "C" int rgbd360_frame_planes(rgbd360_ctx* ctx_, const void* depth, size_t depth_step, int depth_type, int rows, int cols,
                                    int convention, float max_depth_change_factor, float normal_smoothing_size, int min_inliers,
                                    float angular_threshold, float distance_threshold, float max_curvature, int depth_mode,
                                    float* xyz_out, float* normals_out, int32_t* labels_out, rgbd360_plane* planes_out,
                                    int max_planes, int* n_planes_out) {
    F360_ENTER(ctx_);
    return frame_planes_impl(ctx, depth, depth_step, depth_type, rows, cols, convention, max_depth_change_factor, normal_smoothing_size,
                             min_inliers, angular_threshold, distance_threshold, max_curvature, depth_mode, xyz_out, normals_out,
                             labels_out, planes_out, max_planes, n_planes_out, false);
}

extern "C" int rgbd360_frame_planes_dev(rgbd360_ctx* ctx_, const void* depth_dev, size_t depth_step, int depth_type, int rows, int cols,
                                        int convention, float max_depth_change_factor, float normal_smoothing_size, int min_inliers,
                                        float angular_threshold, float distance_threshold, float max_curvature, int depth_mode,
                                        rgbd360_plane* planes_out, int max_planes, int* n_planes_out, const float** xyz_dev,
                                        const float** normals_dev, const int32_t** labels_dev) {
    F360_ENTER(ctx_);
    const int rc = frame_planes_impl(ctx, depth_dev, depth_step, depth_type, rows, cols, convention, max_depth_change_factor,
                                     normal_smoothing_size, min_inliers, angular_threshold, distance_threshold, max_curvature, depth_mode,
                                     nullptr, nullptr, nullptr, planes_out, max_planes, n_planes_out, true);
    if (rc) return rc;
    if (xyz_dev) *xyz_dev = ctx->f_xyz;
    if (normals_dev) *normals_dev = ctx->f_normals;
    if (labels_dev) *labels_dev = ctx->f_label;
    return 0;
}

extern "C" int rgbd360_stitch_sphere(rgbd360_ctx* ctx_, const uint8_t* rgb8, const uint16_t* depth8, int sensor_rows, int sensor_cols,
                                     const float Rt_inv[128], const float K[4], uint8_t* sphere_rgb_out, uint16_t* sphere_depth_out,
                                     int* out_rows, int* out_cols) {
    F360_ENTER(ctx_);
    if (!ctx || !rgb8 || !depth8 || !Rt_inv || !K || !sphere_rgb_out || !sphere_depth_out) return -1;
    if (sensor_rows < 1 || sensor_cols < 1 || sensor_rows > 4096 || sensor_cols > 4096) return fail(ctx, -1, "bad sensor image size");
    hipSetDevice(ctx->p.device);
    f360::StitchArgs a;
    memcpy(a.Rt_inv, Rt_inv, sizeof(a.Rt_inv));
    a.fx = K[0]; a.fy = K[1]; a.cx = K[2]; a.cy = K[3];
    a.sensor_rows = sensor_rows; a.sensor_cols = sensor_cols;
    a.W = sensor_rows * 8;                               // Frame360.h:391
    a.H = (int)(a.W * 0.5 * 60.0 / 180);                 // Frame360.h:392
    if (out_rows) *out_rows = a.H;
    if (out_cols) *out_cols = a.W;
    const float offsetPhi = a.H / 2 - 0.5;               // Frame360.h:1104-1106
    const float offsetTheta = -sensor_rows * 15 / 2 + 0.5;
    const float angle_pixel = 2 * kPI / a.W;
    std::vector<float> tab((size_t)2 * a.H + 2 * a.W);
    for (int r = 0; r < a.H; ++r) {
        const float phi_i = (offsetPhi - r) * angle_pixel;
        tab[r] = sinf(phi_i);
        tab[a.H + r] = cosf(phi_i);
    }
    for (int c = 0; c < a.W; ++c) {
        const float theta_i = (c + offsetTheta) * angle_pixel;
        tab[2 * a.H + c] = sinf(theta_i);
        tab[2 * a.H + a.W + c] = cosf(theta_i);
    }
    const size_t n_in = (size_t)8 * sensor_rows * sensor_cols, n_out = (size_t)a.H * a.W;
    uint8_t *d_rgb = nullptr, *d_out_rgb = nullptr;
    uint16_t *d_depth = nullptr, *d_out_depth = nullptr;
    float* d_tab = nullptr;
    hipError_t e = hipMalloc(&d_rgb, n_in * 3);
    if (e == hipSuccess) e = hipMalloc(&d_depth, n_in * 2);
    if (e == hipSuccess) e = hipMalloc(&d_out_rgb, n_out * 3);
    if (e == hipSuccess) e = hipMalloc(&d_out_depth, n_out * 2);
    if (e == hipSuccess) e = hipMalloc(&d_tab, tab.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(d_rgb, rgb8, n_in * 3, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_depth, depth8, n_in * 2, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(f360::k_stitch_sphere, grid2d(a.H, a.W), dim3(256), 0, ctx->stream, a, d_rgb, d_depth, d_tab, d_tab + a.H,
                           d_tab + 2 * a.H, d_tab + 2 * a.H + a.W, d_out_rgb, d_out_depth);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(sphere_rgb_out, d_out_rgb, n_out * 3, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(sphere_depth_out, d_out_depth, n_out * 2, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(d_rgb); hipFree(d_depth); hipFree(d_out_rgb); hipFree(d_out_depth); hipFree(d_tab);
    HIPC(ctx, e);
    return 0;
}


#ifdef RGBD360_HULL_DBG
extern "C" int rgbd360_debug_hull_stats(unsigned long long* out /* [4096][8] */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(f360::g_hull_dbg), sizeof(unsigned long long) * 4096 * 8) == hipSuccess ? 0 : -1;
}
#endif

