// sequence_engine.h -- lock-step batch of alignments for an odometry sequence (BASELINE.json configs[3]; the per-GPU engine behind
// rgbd360_align360_batch[_dev] and the multi-GPU entry).  Included by rgbd360_api.hip.
//
// A 4-level alignment is ~35 dependent launches, most of them over 32 K - 500 K pixels: pure launch latency on a 256-CU part.
// The engine keeps P pairs ("slots") in flight on ONE stream and gives every launch a slot dimension: one k_eval_b /
// k_solve_b per {level, iteration} serves all P pairs (each slot gated by its own device-resident GNState, exactly like the
// one-pair kernels), and the frame set-up kernels convert / reduce / differentiate the P new frames of a round together.  The
// cost of a round of P alignments is then the launch count of ONE alignment plus P times the bandwidth-bound work.
//
// Slot s owns the contiguous pairs [a_s, b_s) of the sequence (frame reuse inside the span: the source of pair j is the target
// of pair j+1, its planes change role and only its gradient records are built); round r aligns pair a_s + r in every slot that
// still has one (live mask).  Work split, partial rows and summation order per slot are those of the one-pair path
// (RPI.h:4519-4784 semantics unchanged): poses are bit-identical to rgbd360_align360 pair by pair.
#pragma once

namespace {

struct SeqLevel {
    int rows = 0, cols = 0, n = 0;
    float half_nRows = 0.f, angle_res_inv = 0.f;
    float *gray = nullptr, *depth = nullptr;                                    // [P][n] planes of levels >= 1 (level 0 is never stored)
    float4* srcRec = nullptr;                                                   // [P][n]
    F3 *trgP[2] = {nullptr, nullptr}, *trgD[2] = {nullptr, nullptr};            // [P][n] x 2: the records of a frame are built when it
                                                                                // arrives as a source, one round before it is the target
    float *sinT = nullptr, *cosT = nullptr, *sinP = nullptr, *cosP = nullptr;
    float2 *tabT = nullptr, *tabP = nullptr;      // interleaved {sin, cos} (recompute form of the pass)
    bool compact = false;                         // source records of this level are {depth, Isrc} (8 B) and k_eval_b re-forms the point
    int nblocks = 0, chunk = 0;
};

struct SeqEngine {
    rgbd360_params p;
    int P = 0, rows = 0, cols = 0;
    hipStream_t stream = nullptr, up_stream = nullptr;
    hipEvent_t up_ev[2] = {nullptr, nullptr}, conv_ev[2] = {nullptr, nullptr};
    std::vector<SeqLevel> levels;
    GNState* d_states = nullptr;
    GNState* h_states = nullptr;      // pinned
    double* d_partials = nullptr;
    int partials_stride = 0;          // doubles per slot
    uint8_t *stage_rgb[2] = {nullptr, nullptr}, *stage_depth[2] = {nullptr, nullptr};      // [P] frames each (host-frame sequences)
    size_t stage_rgb_frame = 0, stage_depth_frame = 0;
    int tb = 0;                       // which target-record buffer holds the TARGETS of the round
    int max_eval_blocks = 256;
    int libm = 0;                 // the warp in the reference's libm arithmetic (the owning context's rgbd360_set_index_arithmetic)
    int chunk_top = 8, chunk_mid = 4, chunk_l0 = 4;      // {pass, solve} pairs enqueued ahead per level and visit
    std::string err;
};

#define SEQC(E, expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            (E)->err = std::string(#expr) + ": " + hipGetErrorString(e_);               \
            return -(int)e_ - 1000;                                                     \
        }                                                                               \
    } while (0)

void seq_free(SeqEngine* E) {
    if (!E) return;
    hipSetDevice(E->p.device);
    if (E->stream) hipStreamSynchronize(E->stream);
    if (E->up_stream) hipStreamSynchronize(E->up_stream);
    for (SeqLevel& L : E->levels) {
        hipFree(L.gray); hipFree(L.depth); hipFree(L.srcRec);
        for (int k = 0; k < 2; ++k) { hipFree(L.trgP[k]); hipFree(L.trgD[k]); }
        hipFree(L.sinT); hipFree(L.cosT); hipFree(L.sinP); hipFree(L.cosP); hipFree(L.tabT); hipFree(L.tabP);
    }
    hipFree(E->d_states); hipFree(E->d_partials);
    if (E->h_states) hipHostFree(E->h_states);
    for (int k = 0; k < 2; ++k) {
        hipFree(E->stage_rgb[k]); hipFree(E->stage_depth[k]);
        if (E->up_ev[k]) hipEventDestroy(E->up_ev[k]);
        if (E->conv_ev[k]) hipEventDestroy(E->conv_ev[k]);
    }
    if (E->up_stream) hipStreamDestroy(E->up_stream);
    if (E->stream) hipStreamDestroy(E->stream);
    delete E;
}

// Geometry, tables and work split exactly as ensure_levels() builds them for a one-pair context.
int seq_create(const rgbd360_params& p, int P, int rows, int cols, int max_eval_blocks, SeqEngine** out, std::string* err) {
    *out = nullptr;
    if (P < 1 || P > kMaxSlots) { *err = "slots per engine must be in 1..32"; return -1; }
    if (rows < 2 || cols < 8) { *err = "image too small"; return -1; }
    if ((rows >> (p.n_pyr - 1)) < 2 || (cols >> (p.n_pyr - 1)) < 8) {
        *err = "too many pyramid levels for this image size (coarsest level must be >= 2 x 8)";
        return -1;
    }
    if ((long long)rows * cols >= (1ll << 24) || rows >= (1 << 15) || cols >= (1 << 15)) {
        *err = "image too large (the fused pass uses 24-bit index arithmetic: < 16 Mpx)";
        return -1;
    }
    if (hipSetDevice(p.device) != hipSuccess) { *err = "hipSetDevice failed"; return -102; }
    SeqEngine* E = new SeqEngine();
    E->p = p; E->P = P; E->rows = rows; E->cols = cols; E->max_eval_blocks = max_eval_blocks;
    if (const char* e = knobs::debug("RGBD360_SEQ_CHUNKS")) {        // "top,mid,l0" tuning knob
        int a = 0, b = 0, c = 0;
        if (sscanf(e, "%d,%d,%d", &a, &b, &c) == 3 && a >= 1 && a <= 16 && b >= 1 && b <= 16 && c >= 1 && c <= 16) {
            E->chunk_top = a; E->chunk_mid = b; E->chunk_l0 = c;
        }
    }
    auto bad = [&](const char* what) {
        *err = what;
        seq_free(E);
        return -103;
    };
    if (hipStreamCreateWithFlags(&E->stream, hipStreamNonBlocking) != hipSuccess) return bad("cannot create the engine's stream");
    E->levels.resize(p.n_pyr);
    int r = rows, c = cols, max_blocks = 0;
    for (int l = 0; l < p.n_pyr; ++l) {
        SeqLevel& L = E->levels[l];
        L.rows = r; L.cols = c; L.n = r * c;
        const float angle_res = 2 * kPI / c;        // RPI.h:2554
        L.angle_res_inv = 1 / angle_res;            // RPI.h:2555
        L.half_nRows = 0.5 * r - 0.5;               // RPI.h:2557
        const size_t np = (size_t)P * L.n;
        bool ok = true;
        if (l > 0) ok = ok && hipMalloc(&L.gray, np * sizeof(float)) == hipSuccess && hipMalloc(&L.depth, np * sizeof(float)) == hipSuccess;
        ok = ok && hipMalloc(&L.srcRec, np * sizeof(float4)) == hipSuccess;
        for (int k = 0; k < 2; ++k)
            ok = ok && hipMalloc(&L.trgP[k], np * sizeof(F3)) == hipSuccess && hipMalloc(&L.trgD[k], np * sizeof(F3)) == hipSuccess;
        ok = ok && hipMalloc(&L.sinT, c * sizeof(float)) == hipSuccess && hipMalloc(&L.cosT, c * sizeof(float)) == hipSuccess &&
             hipMalloc(&L.sinP, r * sizeof(float)) == hipSuccess && hipMalloc(&L.cosP, r * sizeof(float)) == hipSuccess;
        if (!ok) return bad("out of device memory for the sequence engine");
        std::vector<float> st(c), ct(c), sp(r), cp(r);      // RPI.h:4556-4571 (host libm, as ensure_levels)
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
        ok = hipMemcpy(L.sinT, st.data(), c * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(L.cosT, ct.data(), c * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(L.sinP, sp.data(), r * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(L.cosP, cp.data(), r * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
        if (!ok) return bad("table upload failed");
        {
            std::vector<float2> tt(c), tp(r);
            for (int j = 0; j < c; ++j) tt[j] = make_float2(st[j], ct[j]);
            for (int i = 0; i < r; ++i) tp[i] = make_float2(sp[i], cp[i]);
            ok = hipMalloc(&L.tabT, c * sizeof(float2)) == hipSuccess && hipMalloc(&L.tabP, r * sizeof(float2)) == hipSuccess &&
                 hipMemcpy(L.tabT, tt.data(), c * sizeof(float2), hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(L.tabP, tp.data(), r * sizeof(float2), hipMemcpyHostToDevice) == hipSuccess;
            if (!ok) return bad("table upload failed");
        }
        // The engine's launches serve P pairs at once, so its large levels are fed from HBM whatever the image size: they carry the
        // 8-byte {depth, Isrc} source record and k_eval_b re-forms the point (SrcForm<2>: 32 instead of 40 B/px per pass, 32 instead of
        // 40 B/px written by the set-up).  The latency-bound small levels keep the 16-byte record.  RGBD360_SEQ_RECOMPUTE_MIN_PX moves
        // the bound (0: every level; a huge value: none).  The rig's pinhole records never take this form (rig_dense.h builds its own).
        static const int seq_min_px = [] { const char* e = knobs::product("RGBD360_SEQ_RECOMPUTE_MIN_PX"); return e ? atoi(e) : 256 * 1024; }();
        L.compact = L.n >= seq_min_px;
        int chunk = (L.n + max_eval_blocks - 1) / max_eval_blocks;
        chunk = ((chunk + kEvalThreads - 1) / kEvalThreads) * kEvalThreads;
        L.chunk = chunk;
        L.nblocks = (L.n + chunk - 1) / chunk;
        max_blocks = std::max(max_blocks, L.nblocks);
        r /= 2; c /= 2;
    }
    E->partials_stride = max_blocks * kNumPartials;
    if (hipMalloc(&E->d_partials, (size_t)P * E->partials_stride * sizeof(double)) != hipSuccess ||
        hipMalloc(&E->d_states, (size_t)P * sizeof(GNState)) != hipSuccess ||
        hipMemset(E->d_states, 0, (size_t)P * sizeof(GNState)) != hipSuccess ||
        hipHostMalloc((void**)&E->h_states, (size_t)P * sizeof(GNState), hostwait::kPublishedFlags) != hipSuccess)
        return bad("out of memory for the engine state");
    *out = E;
    return 0;
}

LevelDev seq_level_dev(const SeqLevel& L, int tb) {
    LevelDev d;
    d.rows = L.rows; d.cols = L.cols; d.n = L.n;
    d.half_nRows = L.half_nRows; d.angle_res_inv = L.angle_res_inv;
    d.pi_k = (float)(kPI * (double)L.angle_res_inv);
    d.src = L.srcRec; d.trgP = L.trgP[tb]; d.trgD = L.trgD[tb];
    d.src2 = reinterpret_cast<const float2*>(L.srcRec); d.tabT = L.tabT; d.tabP = L.tabP;
    return d;
}

void seq_launch_eval(SeqEngine* E, int level, int method) {
    const SeqLevel& L = E->levels[level];
    LevelDev lv = seq_level_dev(L, E->tb);
    lv.min_depth = E->p.min_depth; lv.max_depth = E->p.max_depth;
    lv.libm = E->libm;
    const EvalConsts ec = eval_consts(E->p);
    const dim3 g(L.nblocks, E->P), b(kEvalThreadsBatch);
#define LAUNCHB(M, S) hipLaunchKernelGGL((k_eval_b<M, true, S>), g, b, 0, E->stream, E->d_states, lv.src, lv.n, L.chunk, level, L.nblocks, E->d_partials, E->partials_stride, lv, ec)
    if (L.compact) {
        if (method == 0) LAUNCHB(0, 2);
        else if (method == 1) LAUNCHB(1, 2);
        else LAUNCHB(2, 2);
    } else {
        if (method == 0) LAUNCHB(0, 0);
        else if (method == 1) LAUNCHB(1, 0);
        else LAUNCHB(2, 0);
    }
#undef LAUNCHB
}

void seq_launch_solve(SeqEngine* E, int level, int forced = 0) {
    const SeqLevel& L = E->levels[level];
    SolveCfg cfg;
    cfg.level = level; cfg.mode = 0; cfg.forced = forced; cfg.max_iters = E->p.max_iters; cfg.n_pixels = L.n;
    cfg.occ = 0;
    cfg.tol_residual = E->p.tol_residual; cfg.tol_update = E->p.tol_update;
    hipLaunchKernelGGL(k_solve_b, dim3(E->P), dim3(kSolveThreads), 0, E->stream, E->d_states, E->d_partials, E->partials_stride, L.nblocks, cfg);
}

void seq_enqueue_schedule(SeqEngine* E, int pending, bool pending_started, const float* guess, int method, unsigned long long live) {
    const int top = E->p.n_pyr - 1;
    for (int level = pending; level >= 0; --level) {
        if (level == top && !pending_started) {
            Pose16 Pz;
            memcpy(Pz.v, guess, sizeof(Pz.v));
            hipLaunchKernelGGL(k_level_init_b, dim3(E->P), dim3(64), 0, E->stream, E->d_states, Pz, 1, 1, level, live);
        }
        const int n_pairs = (level == top && !pending_started) ? E->chunk_top : (level == 0 ? E->chunk_l0 : E->chunk_mid);
        for (int k = 0; k < n_pairs; ++k) {
            seq_launch_eval(E, level, method);
            seq_launch_solve(E, level);
        }
    }
}

// One frame per live slot through the fused set-up (k_frame_level_b, one launch per pyramid level): source records for the slots
// of src_mask, target records (into buffer trg_buf) for those of trg_mask, next-level planes for all.
void seq_frame_setup(SeqEngine* E, const FramePtrs& fp, size_t rgb_step, size_t depth_step, int depth_type, unsigned long long live,
                     unsigned long long src_mask, unsigned long long trg_mask, int trg_buf) {
    for (int l = 0; l < E->p.n_pyr; ++l) {
        const SeqLevel& L = E->levels[l];
        FrameLevelArgs A;
        memset(&A, 0, sizeof(A));
        A.rows = L.rows; A.cols = L.cols;
        if (l + 1 < E->p.n_pyr) {
            const SeqLevel& N = E->levels[l + 1];
            A.drows = N.rows; A.dcols = N.cols;
            A.gray_next = N.gray; A.depth_next = N.depth;
        }
        A.seam = E->p.mask_seams ? L.cols / 8 : 0;
        A.depth_type = depth_type;
        A.rgb_step = rgb_step; A.depth_step = depth_step;
        A.gray_in = L.gray; A.depth_in = L.depth;
        A.src_rec = L.srcRec; A.trg_p = L.trgP[trg_buf]; A.trg_d = L.trgD[trg_buf];
        A.sin_theta = L.sinT; A.cos_theta = L.cosT; A.sin_phi = L.sinP; A.cos_phi = L.cosP;
        A.min_depth = E->p.min_depth; A.max_depth = E->p.max_depth;
        A.compact_src = L.compact ? 1 : 0;
        A.live_mask = live; A.src_mask = src_mask; A.trg_mask = trg_mask;
        const dim3 g((L.cols + kFsTW - 1) / kFsTW, (L.rows + kFsTH - 1) / kFsTH, E->P);
        if (l == 0) hipLaunchKernelGGL((k_frame_level_b<true>), g, dim3(256), 0, E->stream, A, fp);
        else hipLaunchKernelGGL((k_frame_level_b<false>), g, dim3(256), 0, E->stream, A, fp);
    }
}

int seq_ensure_stage(SeqEngine* E, int depth_type) {
    const size_t fr = (size_t)E->rows * E->cols * 3, fd = (size_t)E->rows * E->cols * (depth_type == 0 ? 2 : 4);
    if (E->stage_rgb_frame == fr && E->stage_depth_frame == fd && E->up_stream) return 0;
    if (!E->up_stream) {
        SEQC(E, hipStreamCreateWithFlags(&E->up_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            SEQC(E, hipEventCreateWithFlags(&E->up_ev[k], hipEventDisableTiming));
            SEQC(E, hipEventCreateWithFlags(&E->conv_ev[k], hipEventDisableTiming));
        }
    }
    SEQC(E, hipStreamSynchronize(E->stream));
    SEQC(E, hipStreamSynchronize(E->up_stream));
    for (int k = 0; k < 2; ++k) {
        hipFree(E->stage_rgb[k]); hipFree(E->stage_depth[k]);
        E->stage_rgb[k] = E->stage_depth[k] = nullptr;
        SEQC(E, hipMalloc(&E->stage_rgb[k], fr * E->P));
        SEQC(E, hipMalloc(&E->stage_depth[k], fd * E->P));
    }
    E->stage_rgb_frame = fr; E->stage_depth_frame = fd;
    return 0;
}

void result_from_state(const GNState& S, int n_pyr, int occ, float pose_out[16], rgbd360_result* res) {
    rgbd360_result R;
    memset(&R, 0, sizeof(R));
    for (int l = 0; l < n_pyr && l < 8; ++l) R.iters[l] = S.iters[l];
    memcpy(pose_out, S.pose, sizeof(float) * 16);
    R.status = S.status;
    memcpy(R.hessian, S.Hused, sizeof(R.hessian));
    memcpy(R.gradient, S.gused, sizeof(R.gradient));
    R.sso = S.used_npix ? (float)S.used_nvis / (float)S.used_npix : 0.f;
    const double nv = (double)(S.acc_np + S.acc_nd);
    R.err_final = nv > 0 ? sqrt((S.acc_e2p + S.acc_e2d) / nv) : 0.0;
    if (occ != 0)       // avPhotoResidual + avDepthResidual (RPI.h:3358-3366, 3848-3855)
        R.err_final = (S.acc_np > 0 ? sqrt(S.acc_e2p / (double)S.acc_np) : 0.0) + (S.acc_nd > 0 ? sqrt(S.acc_e2d / (double)S.acc_nd) : 0.0);
    R.rms_photo = S.acc_np > 0 ? sqrt(S.acc_e2p / (double)S.acc_np) : 0.0;
    R.rms_depth = S.acc_nd > 0 ? sqrt(S.acc_e2d / (double)S.acc_nd) : 0.0;
    if (res) *res = R;
}

// Slot s aligns the pairs [a[s], b[s]) of the sequence rgb[] / depth[] (global frame indices; pair j = frames j, j+1).
// (seq_run below wraps this body with the one exit that matters: whatever the outcome, no copy from the caller's images and no
// launch of this call is still in flight when the C call returns.)
int seq_run_body(SeqEngine* E, int n_slots, const int* a, const int* b, const uint8_t* const* rgb, size_t rgb_step, const void* const* depth,
            size_t depth_step, int depth_type, const float* guess, int method, bool on_device, float* poses_out, rgbd360_result* results_out) {
    hipSetDevice(E->p.device);
    const int P = E->P;
    if (n_slots < 1 || n_slots > P) { E->err = "bad slot count"; return -1; }
    int rounds = 0;
    for (int s = 0; s < n_slots; ++s) rounds = std::max(rounds, b[s] - a[s]);
    if (rounds == 0) return 0;
    const size_t dpx = depth_type == 0 ? 2 : 4;
    if (!on_device) {
        const int rc = seq_ensure_stage(E, depth_type);
        if (rc) return rc;
    }
    auto live_of = [&](int r) {        // slots that still have a pair in round r
        unsigned long long m = 0;
        for (int s = 0; s < n_slots; ++s)
            if (a[s] + r < b[s]) m |= 1ull << s;
        return m;
    };
    // frames of a round: which = 0 the targets of round 0 (frame a[s]), 1 the sources of round r (frame a[s] + r + 1)
    auto frame_of = [&](int s, int r, int which) { return which == 0 ? a[s] : a[s] + r + 1; };
    // host frames travel through staging parity `par`: upload on the copy stream (after the parity's previous frames were converted)
    auto upload = [&](int r, int which, int par, unsigned long long live) -> int {
        SEQC(E, hipStreamWaitEvent(E->up_stream, E->conv_ev[par], 0));
        for (int s = 0; s < n_slots; ++s) {
            if (!((live >> s) & 1ull)) continue;
            const int f = frame_of(s, r, which);
            SEQC(E, hipMemcpy2DAsync(E->stage_rgb[par] + (size_t)s * E->stage_rgb_frame, (size_t)E->cols * 3, rgb[f], rgb_step, (size_t)E->cols * 3,
                                     E->rows, hipMemcpyHostToDevice, E->up_stream));
            SEQC(E, hipMemcpy2DAsync(E->stage_depth[par] + (size_t)s * E->stage_depth_frame, (size_t)E->cols * dpx, depth[f], depth_step,
                                     (size_t)E->cols * dpx, E->rows, hipMemcpyHostToDevice, E->up_stream));
        }
        SEQC(E, hipEventRecord(E->up_ev[par], E->up_stream));
        return 0;
    };
    auto frame_ptrs = [&](int r, int which, int par, unsigned long long live, FramePtrs* fp, size_t* rs, size_t* ds) {
        memset(fp, 0, sizeof(*fp));
        for (int s = 0; s < n_slots; ++s) {
            if (!((live >> s) & 1ull)) continue;
            if (on_device) {
                const int f = frame_of(s, r, which);
                fp->rgb[s] = rgb[f];
                fp->depth[s] = depth[f];
            } else {
                fp->rgb[s] = E->stage_rgb[par] + (size_t)s * E->stage_rgb_frame;
                fp->depth[s] = E->stage_depth[par] + (size_t)s * E->stage_depth_frame;
            }
        }
        *rs = on_device ? rgb_step : (size_t)E->cols * 3;
        *ds = on_device ? depth_step : (size_t)E->cols * dpx;
    };
    int rc = 0;
    int par = 0;      // staging parity of the NEXT conversion
    if (!on_device) {
        if ((rc = upload(0, 0, 0, live_of(0))) != 0) return rc;      // targets of round 0 -> parity 0
        if ((rc = upload(0, 1, 1, live_of(0))) != 0) return rc;      // sources of round 0 -> parity 1
    }
    for (int r = 0; r < rounds; ++r) {
        const unsigned long long live = live_of(r);
        FramePtrs fp;
        size_t rs, ds;
        if (r == 0) {
            // first targets: frames a[s], target records only, into buffer tb
            frame_ptrs(0, 0, par, live, &fp, &rs, &ds);
            if (!on_device) SEQC(E, hipStreamWaitEvent(E->stream, E->up_ev[par], 0));
            seq_frame_setup(E, fp, rs, ds, depth_type, live, 0ull, live, E->tb);
            if (!on_device) { SEQC(E, hipEventRecord(E->conv_ev[par], E->stream)); par ^= 1; }
        }
        // sources of this round; they are the targets of the next one in every slot that has another pair: their target
        // records go into the other buffer now, while the tile is in LDS anyway
        frame_ptrs(r, 1, par, live, &fp, &rs, &ds);
        if (!on_device) SEQC(E, hipStreamWaitEvent(E->stream, E->up_ev[par], 0));
        seq_frame_setup(E, fp, rs, ds, depth_type, live, live, live & live_of(r + 1), E->tb ^ 1);
        if (!on_device) {
            SEQC(E, hipEventRecord(E->conv_ev[par], E->stream));
            par ^= 1;
        }
        seq_enqueue_schedule(E, E->p.n_pyr - 1, false, guess, method, live);
        SEQC(E, hipGetLastError());
        // next round's sources travel while this round is being aligned (a pageable-memory copy keeps the host in the call, so
        // it is issued only now that the round's launches are queued)
        if (!on_device && r + 1 < rounds && (rc = upload(r + 1, 1, par, live_of(r + 1))) != 0) return rc;
        for (int round = 0;; ++round) {
            SEQC(E, hipMemcpyAsync(E->h_states, E->d_states, (size_t)n_slots * sizeof(GNState), hipMemcpyDeviceToHost, E->stream));
            SEQC(E, hipStreamSynchronize(E->stream));
            int pending = -1;
            for (int s = 0; s < n_slots; ++s) {
                if (!((live >> s) & 1ull)) continue;
                const GNState& S = E->h_states[s];
                if (S.status != 0 || (S.level_active == 0 && S.done)) continue;
                if (S.done) { E->err = "alignment schedule stalled between levels"; return -6; }
                pending = std::max(pending, S.level_active);
            }
            if (pending < 0) break;
            if (round > (E->p.max_iters + 4) * E->p.n_pyr) { E->err = "alignment loop did not terminate"; return -6; }
            seq_enqueue_schedule(E, pending, true, guess, method, live);      // the stalled level gets another chunk, then the finer ones
            SEQC(E, hipGetLastError());
        }
        for (int s = 0; s < n_slots; ++s) {
            if (!((live >> s) & 1ull)) continue;
            const int j = a[s] + r;
            result_from_state(E->h_states[s], E->p.n_pyr, 0, poses_out + (size_t)16 * j, results_out ? &results_out[j] : nullptr);
        }
        E->tb ^= 1;      // this round's sources are the next round's targets
    }
    if (E->up_stream) SEQC(E, hipStreamSynchronize(E->up_stream));      // no upload may outlive the caller's buffers
    return 0;
}

int seq_run(SeqEngine* E, int n_slots, const int* a, const int* b, const uint8_t* const* rgb, size_t rgb_step, const void* const* depth,
            size_t depth_step, int depth_type, const float* guess, int method, bool on_device, float* poses_out, rgbd360_result* results_out) {
    const int rc = seq_run_body(E, n_slots, a, b, rgb, rgb_step, depth, depth_step, depth_type, guess, method, on_device, poses_out, results_out);
    if (rc != 0) {
        // an error return left the body early: hipMemcpy2DAsync copies from the caller's host images may still be queued on the copy
        // stream, staged frames unconverted and event waits pending on the engine's stream.  Drain both (errors here change nothing:
        // the call already failed) so that the caller may free its buffers and the kept engine starts its next call clean.
        if (E->up_stream) (void)hipStreamSynchronize(E->up_stream);
        if (E->stream) (void)hipStreamSynchronize(E->stream);
    }
    return rc;
}

}  // namespace
