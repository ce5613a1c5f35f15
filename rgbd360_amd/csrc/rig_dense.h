// rig_dense.h -- dense registration of two frames of the 8-sensor rig: RegisterRGBD360::RegisterDensePhotoICP
// (RegisterRGBD360.h:344-520) over calcPhotoICPError_robot (RPI.h:4905-5076) and calcHessianGradient_robot (RPI.h:5083-5407).
// SURVEY.md 8f rank 3.  Included by rgbd360_api.hip after sequence_engine.h (shares its fused frame set-up).
//
// The unknown is the RIG's relative pose T (p_rig1 = T p_rig2); sensor s sees it through its extrinsic Rt_s (sensor -> rig).  One
// fused pass evaluates, for all sensors in ONE launch (blockIdx.y = sensor), the error sums of calcPhotoICPError_robot (every
// visible pixel, no saliency test) and the normal equations of calcHessianGradient_robot (saliency-gated rows) at one pose;
// the rows are accumulated directly in the rig's left-perturbation coordinates, so the 8 per-sensor H, g just add up.
//
// The reference function cannot be followed to the letter -- as written it never accepts a step and reads an uninitialised
// Jacobian row.  SURVEY.md asks for it "with the reference's bugs fixed"; the three fixes (A: new_error at the candidate pose,
// B: jacobianRt_z = row 2 of the transform Jacobian, C: depth residual against the TRANSFORMED point's depth) are documented in
// oracle/photo_icp_ref.cpp, which restates the same fixed function line by line and is this file's checker.
//
// Device arithmetic definition of the warp (the oracle's math_mode 1 repeats it): q = (T Rt_s) p and P' = Rt_s^-1 q with fused
// multiply-adds, correctly rounded 1/Z', column = round(fma(X' fx, 1/Z', ox)), row likewise, round = floor(x + 0.5).
// Row algebra: jacobianT36 = R_s^-1 [I | -skew(q)], so a camera-frame row vector a contributes (b, q x b) with b = R_s a.
#pragma once

namespace r360 {

constexpr int kMaxRigSensors = 8;      // NUM_ASUS_SENSORS
struct RigPoses {
    float M[kMaxRigSensors][12];       // rows of (T * Rt_s):   q  = M p      (r00 r01 r02 tx | r10 ... | r20 ...)
    float Ri[kMaxRigSensors][12];      // rows of Rt_s^-1:      P' = Ri q
};

__device__ __forceinline__ void xform12(const float* m, float x, float y, float z, float& X, float& Y, float& Z) {
    X = fmaf(m[2], z, fmaf(m[1], y, fmaf(m[0], x, m[3])));
    Y = fmaf(m[6], z, fmaf(m[5], y, fmaf(m[4], x, m[7])));
    Z = fmaf(m[10], z, fmaf(m[9], y, fmaf(m[8], x, m[11])));
}

template <int METHOD>
__global__ __launch_bounds__(kEvalThreads) void k_eval_rig(const float4* __restrict__ src, const F3* __restrict__ trgP,
                                                            const F3* __restrict__ trgD, int rows, int cols, int n, PinK K, EvalConsts ec,
                                                            RigPoses poses, double* __restrict__ partials, int partials_stride, int chunk, float sal_thr) {
    const int b = blockIdx.x, s = blockIdx.y;
    const int base = b * chunk;
    const int end = min(base + chunk, n);
    src += (size_t)s * n; trgP += (size_t)s * n; trgD += (size_t)s * n;
    const float* Mq = poses.M[s];
    const float* Ri = poses.Ri[s];

    EvalAcc A;
#pragma unroll
    for (int k = 0; k < 27; ++k) A.acc[k] = 0.f;
    A.e2p = A.e2d = 0.f;
    A.nP = A.nD = A.nVis = 0;

    const int n_steps = (end - base + kEvalThreads - 1) / kEvalThreads;      // wave-uniform: the ballots count whole waves
    for (int k = 0; k < n_steps; ++k) {
        const int i = base + k * kEvalThreads + (int)threadIdx.x;
        const bool in_range = i < end;
        const float4 p = src[in_range ? i : n - 1];
        float qx, qy, qz, X, Y, Z;
        xform12(Mq, p.x, p.y, p.z, qx, qy, qz);
        xform12(Ri, qx, qy, qz, X, Y, Z);
        const float iz = rcp_rn(Z);
        const float tc = fmaf(X * K.fx, iz, K.ox);
        const float tr = fmaf(Y * K.fy, iz, K.oy);
        const bool sane = (fabsf(tr) < 1e9f) && (fabsf(tc) < 1e9f);
        const int ri = round_index(sane ? tr : -1.f), ci = round_index(sane ? tc : -1.f);
        bool vis = sane && ((unsigned)ri < (unsigned)rows) && ((unsigned)ci < (unsigned)cols) && in_range && (p.x != kInvalidPoint);
        if (sal_thr >= 0.f) {      // uniform: bUseSalientPixels (RPI.h:4930-5003, 5121-5262) -- both passes run over vSalientPixels only, the
            const F3 ts = trgP[in_range ? i : 0];      // interior pixels whose TARGET gray gradient exceeds thresSaliency, used as source indices
            vis = vis && (fabsf(ts.b) > sal_thr || fabsf(ts.c) > sal_thr);
        }
        const unsigned ti = vis ? (unsigned)(ri * cols + ci) : 0u;
        F3 tp = {0.f, 0.f, 0.f}, td = {0.f, 0.f, 0.f};
        if (METHOD != 1) tp = trgP[ti];
        if (METHOD != 0) td = trgD[ti];
        const float depth2 = td.a;
        const bool sal_p = !(fabsf(tp.b) < ec.thr_photo && fabsf(tp.c) < ec.thr_photo);
        const bool sal_d = !(fabsf(td.b) < ec.thr_depth && fabsf(td.c) < ec.thr_depth);
        const bool fin_d = METHOD != 0 && isfinite(depth2);
        const float iz2 = iz * iz;
        // b = R_s a: R_s = (Rt_s^-1 rotation)^T, i.e. b_i = sum_j Ri[j][i] a_j
        auto to_rig = [&](float ax, float ay, float az, float& bx, float& by, float& bz) {
#pragma clang fp contract(fast)
            bx = Ri[0] * ax + Ri[4] * ay + Ri[8] * az;
            by = Ri[1] * ax + Ri[5] * ay + Ri[9] * az;
            bz = Ri[2] * ax + Ri[6] * ay + Ri[10] * az;
        };
        if (METHOD != 1) {
            A.nP += ballot_count(vis);                                             // calcPhotoICPError_robot: no saliency test
            // calcHessianGradient_robot: a non-salient intensity gradient skips the pixel (RPI.h:5331-5332); a finite target depth with a
            // flat depth gradient skips it too, photometric row included (RPI.h:5352-5353)
            const bool row_on = vis && sal_p && (METHOD == 0 || !fin_d || sal_d);
            A.nVis += ballot_count(row_on);
            if (vis) {
#pragma clang fp contract(fast)
                const float photoDiff = tp.a - p.w;
                const float wpf = weight_huber_fast(photoDiff, ec.sigma_photo) * ec.sigma_photo_inv_f;
                const float res = wpf * photoDiff;
                A.e2p += res * res;
                if (row_on) {
                    const float wgx = wpf * tp.b * K.fx, wgy = wpf * tp.c * K.fy;
                    float bx, by, bz;
                    to_rig(wgx * iz, wgy * iz, -(wgx * X + wgy * Y) * iz2, bx, by, bz);
                    accumulate_row(A, bx, by, bz, qx, qy, qz, res);
                }
            }
        }
        if (METHOD != 0) {
            const bool err_on = vis && fin_d;
            A.nD += ballot_count(err_on);
            const bool row_on = err_on && sal_d && (METHOD == 1 || sal_p);
            A.nVis += ballot_count(row_on);
            if (err_on) {
#pragma clang fp contract(fast)
                const float depthDiff = depth2 - Z;                                 // FIX C: the transformed point's depth
                const float sd = ec.sigma_depth * Z;
                const float wd = weight_huber_fast(depthDiff, sd) * fast_rcp(sd);
                const float res = wd * depthDiff;
                A.e2d += res * res;
                if (row_on) {
                    const float gx = td.b * K.fx, gy = td.c * K.fy;
                    float bx, by, bz;
                    to_rig(wd * (gx * iz), wd * (gy * iz), wd * (-(gx * X + gy * Y) * iz2 - 1.f), bx, by, bz);      // FIX B: - jacobianT36.row(2)
                    accumulate_row(A, bx, by, bz, qx, qy, qz, res);
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
        partials[(size_t)s * partials_stride + (size_t)b * kNumPartials + threadIdx.x] = v;
    }
}

// Per-sensor totals of the partial rows, fixed order, written straight into pinned host memory: block s -> out[s][32].
// The block that finishes last stores the host's sequence tag (host_wait.h): no tag kernel behind this one.
__global__ __launch_bounds__(256) void k_rig_reduce(const double* __restrict__ partials, int partials_stride, int nb, double* __restrict__ out,
                                                    unsigned* __restrict__ ticket, unsigned* __restrict__ tag, unsigned seq) {
    __shared__ double red[8][kNumPartials];
    const int s = blockIdx.x, v = threadIdx.x & 31, q = threadIdx.x >> 5;
    double acc = 0.0;
    for (int b = q; b < nb; b += 8) acc += partials[(size_t)s * partials_stride + (size_t)b * kNumPartials + v];
    red[q][v] = acc;
    __syncthreads();
    if (threadIdx.x < kNumPartials) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
        out[(size_t)s * kNumPartials + threadIdx.x] = t;
        __threadfence_system();        // this block's totals (written by this wave alone) are on their way to the host before it takes its ticket
    }
    __syncthreads();
    // acq_rel on the ticket: the block that draws the last number synchronises with every earlier block's release, so their host
    // writes (fenced above) are ordered before the tag it stores next
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
        __threadfence_system();
        *ticket = 0u;
        __hip_atomic_store(tag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace r360

struct rgbd360_rig {
    rgbd360_params p;
    int S = 0, rows = 0, cols = 0;
    float cam[4] = {0, 0, 0, 0};
    float Rt[r360::kMaxRigSensors][16], Rt_inv[r360::kMaxRigSensors][16];
    SeqEngine* E = nullptr;           // buffers + fused set-up of S "slots" (one per sensor), created at the first frame
    double* h_tot = nullptr;          // pinned, [S][32]
    hostwait::SpinTag tag;
    unsigned* d_ticket = nullptr;     // device counter of k_rig_reduce's blocks
    bool have_src = false, have_trg = false;
    float sal_thr = -1.f;             // useSaliency(true) on the per-sensor objects: thresSaliency (RPI.h:217); < 0 = off
    std::string err;
};

namespace {

int rfail(rgbd360_rig* R, int code, const std::string& msg) {
    R->err = msg;
    return code;
}

void rigid_inverse(const float* M, float* Inv) {       // [R | t]^-1 = [R^T | -R^T t], float, the oracle's order
    for (int k = 0; k < 16; ++k) Inv[k] = 0.f;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Inv[j * 4 + i] = M[i * 4 + j];
    for (int i = 0; i < 3; ++i) Inv[12 + i] = -((Inv[0 * 4 + i] * M[12] + Inv[1 * 4 + i] * M[13]) + Inv[2 * 4 + i] * M[14]);
    Inv[15] = 1.f;
}

PinK rig_level_K(const rgbd360_rig* R, int level) {      // RPI.h:4916-4920
    const float scaleFactor = 1.0 / pow(2, level);
    return {R->cam[0] * scaleFactor, R->cam[1] * scaleFactor, R->cam[2] * scaleFactor, R->cam[3] * scaleFactor};
}

// the 8 sensor images of one frame -> pyramids + records of every level (fused set-up, 1 launch per level for all sensors)
int rig_set_frames(rgbd360_rig* R, bool target, const uint8_t* const* rgb, size_t rgb_step, const void* const* depth, size_t depth_step,
                   int depth_type, int rows, int cols) {
    if (!rgb || !depth) return rfail(R, -1, "null pointer");
    if (depth_type != 0 && depth_type != 1) return rfail(R, -1, "depth_type must be 0 (u16 mm) or 1 (f32 m)");
    for (int s = 0; s < R->S; ++s)
        if (!rgb[s] || !depth[s]) return rfail(R, -1, "null sensor image");
    hipSetDevice(R->p.device);
    if (!R->E || R->rows != rows || R->cols != cols) {
        seq_free(R->E);
        R->E = nullptr;
        R->have_src = R->have_trg = false;
        std::string err;
        const int rc = seq_create(R->p, R->S, rows, cols, 256, &R->E, &err);
        if (rc) return rfail(R, rc, err);
        R->rows = rows; R->cols = cols;
    }
    SeqEngine* E = R->E;
    int rc = seq_ensure_stage(E, depth_type);
    if (rc) return rfail(R, rc, E->err);
    const size_t dpx = depth_type == 0 ? 2 : 4;
    FramePtrs fp;
    memset(&fp, 0, sizeof(fp));
    for (int s = 0; s < R->S; ++s) {
        hipError_t e = hipMemcpy2DAsync(E->stage_rgb[0] + (size_t)s * E->stage_rgb_frame, (size_t)cols * 3, rgb[s], rgb_step, (size_t)cols * 3, rows,
                                        hipMemcpyHostToDevice, E->stream);
        if (e == hipSuccess)
            e = hipMemcpy2DAsync(E->stage_depth[0] + (size_t)s * E->stage_depth_frame, (size_t)cols * dpx, depth[s], depth_step, (size_t)cols * dpx,
                                 rows, hipMemcpyHostToDevice, E->stream);
        if (e != hipSuccess) return rfail(R, -(int)e - 1000, hipGetErrorString(e));
        fp.rgb[s] = E->stage_rgb[0] + (size_t)s * E->stage_rgb_frame;
        fp.depth[s] = E->stage_depth[0] + (size_t)s * E->stage_depth_frame;
    }
    const unsigned long long live = (1ull << R->S) - 1ull;
    for (int l = 0; l < R->p.n_pyr; ++l) {
        const SeqLevel& L = E->levels[l];
        FrameLevelArgs A;
        memset(&A, 0, sizeof(A));
        A.rows = L.rows; A.cols = L.cols;
        if (l + 1 < R->p.n_pyr) {
            const SeqLevel& N = E->levels[l + 1];
            A.drows = N.rows; A.dcols = N.cols;
            A.gray_next = N.gray; A.depth_next = N.depth;
        }
        A.seam = 0;                                      // no seam mask on a pinhole sensor
        A.depth_type = depth_type;
        A.rgb_step = (size_t)cols * 3; A.depth_step = (size_t)cols * dpx;
        A.gray_in = L.gray; A.depth_in = L.depth;
        A.src_rec = L.srcRec; A.trg_p = L.trgP[0]; A.trg_d = L.trgD[0];
        A.min_depth = R->p.min_depth; A.max_depth = R->p.max_depth;
        A.live_mask = live; A.src_mask = target ? 0ull : live; A.trg_mask = target ? live : 0ull;
        A.pinhole = 1;
        const PinK K = rig_level_K(R, l);
        A.pin_ox = K.ox; A.pin_oy = K.oy;
        A.pin_inv_fx = 1. / K.fx; A.pin_inv_fy = 1. / K.fy;      // RPI.h:4921-4922 (float = double quotient, as pin_prepare_level)
        const dim3 g((L.cols + kFsTW - 1) / kFsTW, (L.rows + kFsTH - 1) / kFsTH, R->S);
        if (l == 0) hipLaunchKernelGGL((k_frame_level_b<true>), g, dim3(256), 0, E->stream, A, fp);
        else hipLaunchKernelGGL((k_frame_level_b<false>), g, dim3(256), 0, E->stream, A, fp);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(E->stream);      // the caller may reuse its host images
    if (e != hipSuccess) return rfail(R, -(int)e - 1000, hipGetErrorString(e));
    if (target) R->have_trg = true; else R->have_src = true;
    return 0;
}

struct RigSums {
    double e2p = 0, e2d = 0;
    long long np = 0, nd = 0, rows = 0;
    float H[36], g[6];
    double H64[36], g64[6];
    double error() const { return e2p + e2d; }      // calcPhotoICPError_robot returns error2, the plain sum
};

// one fused pass over all sensors at rig pose T; per-sensor totals are cast to float and added in sensor order like
// `Hessian += alignSensorID[sensor_id].getHessian()` (RegisterRGBD360.h:435-440)
int rig_eval(rgbd360_rig* R, int level, const float* T, int method, RigSums* out) {
    SeqEngine* E = R->E;
    const SeqLevel& L = E->levels[level];
    RigPoses P;
    for (int s = 0; s < R->S; ++s) {
        float M[16];
        gn::mat4_mul(T, R->Rt[s], M);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) {
                P.M[s][4 * r + c] = M[c * 4 + r];
                P.Ri[s][4 * r + c] = R->Rt_inv[s][c * 4 + r];
            }
    }
    const PinK K = rig_level_K(R, level);
    const EvalConsts ec = eval_consts(R->p);
    const dim3 g(L.nblocks, R->S), b(kEvalThreads);
#define LAUNCHR(Mth) hipLaunchKernelGGL((k_eval_rig<Mth>), g, b, 0, E->stream, L.srcRec, L.trgP[0], L.trgD[0], L.rows, L.cols, L.n, K, ec, P, E->d_partials, E->partials_stride, L.chunk, R->sal_thr)
    if (method == 0) LAUNCHR(0);
    else if (method == 1) LAUNCHR(1);
    else LAUNCHR(2);
#undef LAUNCHR
    hipLaunchKernelGGL(k_rig_reduce, dim3(R->S), dim3(256), 0, E->stream, E->d_partials, E->partials_stride, L.nblocks, R->h_tot, R->d_ticket,
                       R->tag.h, ++R->tag.seq);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hostwait::wait(R->tag, E->stream);      // (spin on a pinned tag: one round trip per LM evaluation)
    if (e != hipSuccess) return rfail(R, -(int)e - 1000, hipGetErrorString(e));
    RigSums S;
    memset(S.H, 0, sizeof(S.H)); memset(S.g, 0, sizeof(S.g));
    memset(S.H64, 0, sizeof(S.H64)); memset(S.g64, 0, sizeof(S.g64));
    for (int s = 0; s < R->S; ++s) {
        const double* tot = R->h_tot + (size_t)s * kNumPartials;
        S.e2p += tot[P_E2P]; S.e2d += tot[P_E2D];
        S.np += (long long)tot[P_NP]; S.nd += (long long)tot[P_ND]; S.rows += (long long)tot[P_NVIS];
        int k = 0;
        for (int a = 0; a < 6; ++a)
            for (int c = a; c < 6; ++c, ++k) {
                const float v = (float)tot[P_H + k];
                S.H[c * 6 + a] += v;
                if (c != a) S.H[a * 6 + c] += v;
                S.H64[c * 6 + a] += tot[P_H + k];
                if (c != a) S.H64[a * 6 + c] += tot[P_H + k];
            }
        for (int a = 0; a < 6; ++a) {
            S.g[a] += (float)tot[P_G + a];
            S.g64[a] += tot[P_G + a];
        }
    }
    *out = S;
    return 0;
}

bool rig_lm_update(const float* H, const float* g, float lambda, const float* pose, float* pose_tmp, float* update) {
    float M[36], inv[36];
    for (int k = 0; k < 36; ++k) M[k] = H[k];
    for (int i = 0; i < 6; ++i) M[i * 6 + i] = H[i * 6 + i] + lambda * H[i * 6 + i];
    if (!gn::inverse6(M, inv)) return false;
    for (int r = 0; r < 6; ++r) {
        float s = 0.f;
        for (int c = 0; c < 6; ++c) s += (-inv[c * 6 + r]) * g[c];
        update[r] = s;
    }
    double ud[6], Ex[16];
    for (int i = 0; i < 6; ++i) ud[i] = (double)update[i];
    gn::se3_exp(ud, Ex);                                 // CPose3D::exp(update) -- the full exponential   RegisterRGBD360.h:455
    float Ef[16];
    for (int k = 0; k < 16; ++k) Ef[k] = (float)Ex[k];
    gn::mat4_mul(Ef, pose, pose_tmp);
    return true;
}

}  // namespace

extern "C" {

void rgbd360_rig_destroy(rgbd360_rig* R) {
    if (!R) return;
    hipSetDevice(R->p.device);
    seq_free(R->E);
    if (R->h_tot) hipHostFree(R->h_tot);
    hostwait::spin_tag_free(&R->tag);
    if (R->d_ticket) hipFree(R->d_ticket);
    delete R;
}

int rgbd360_rig_create(const rgbd360_params* p, int n_sensors, const float* Rt, float fx, float fy, float ox, float oy, rgbd360_rig** out) {
    if (!p || !out || !Rt || n_sensors < 1 || n_sensors > kMaxRigSensors) return -1;
    *out = nullptr;
    if (p->n_pyr < 1 || p->n_pyr > 8) return -1;
    if (!(fx > 0.f) || !(fy > 0.f)) return -1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return -100;      // no HIP device: no fallback
    if (p->device < 0 || p->device >= ndev) return -101;
    if (hipSetDevice(p->device) != hipSuccess) return -102;
    rgbd360_rig* R = new rgbd360_rig();
    R->p = *p;
    R->p.mask_seams = 0;
    R->S = n_sensors;
    R->cam[0] = fx; R->cam[1] = fy; R->cam[2] = ox; R->cam[3] = oy;
    for (int s = 0; s < n_sensors; ++s) {
        memcpy(R->Rt[s], Rt + 16 * s, sizeof(float) * 16);
        rigid_inverse(R->Rt[s], R->Rt_inv[s]);
    }
    if (hipHostMalloc((void**)&R->h_tot, sizeof(double) * kNumPartials * kMaxRigSensors, hostwait::kPublishedFlags) != hipSuccess ||
        hostwait::spin_tag_init(&R->tag) != hipSuccess || hipMalloc(&R->d_ticket, sizeof(unsigned)) != hipSuccess ||
        hipMemset(R->d_ticket, 0, sizeof(unsigned)) != hipSuccess) {
        if (R->h_tot) hipHostFree(R->h_tot);
        hostwait::spin_tag_free(&R->tag);
        if (R->d_ticket) hipFree(R->d_ticket);
        delete R;
        return -103;
    }
    *out = R;
    return 0;
}

const char* rgbd360_rig_last_error(rgbd360_rig* R) { return R ? R->err.c_str() : "null handle"; }

int rgbd360_rig_use_saliency(rgbd360_rig* R, int on, float thres_saliency) {
    if (!R) return -1;
    if (on && !(thres_saliency >= 0.f)) return rfail(R, -1, "thres_saliency must be >= 0");
    R->sal_thr = on ? thres_saliency : -1.f;
    return 0;
}

int rgbd360_rig_set_target(rgbd360_rig* R, const uint8_t* const* rgb, size_t rgb_step, const void* const* depth, size_t depth_step,
                           int depth_type, int rows, int cols) {
    return R ? rig_set_frames(R, true, rgb, rgb_step, depth, depth_step, depth_type, rows, cols) : -1;
}
int rgbd360_rig_set_source(rgbd360_rig* R, const uint8_t* const* rgb, size_t rgb_step, const void* const* depth, size_t depth_step,
                           int depth_type, int rows, int cols) {
    return R ? rig_set_frames(R, false, rgb, rgb_step, depth, depth_step, depth_type, rows, cols) : -1;
}

int rgbd360_rig_eval(rgbd360_rig* R, int level, const float pose[16], int method, double err2_split[2], long long n_split[2], float H[36],
                     float g[6], double H64[36], double g64[6], long long* n_rows) {
    if (!R) return -1;
    if (!R->have_src || !R->have_trg) return rfail(R, -2, "rgbd360_rig_set_target and _set_source must be called first");
    if (level < 0 || level >= R->p.n_pyr) return rfail(R, -3, "bad pyramid level");
    if (method < 0 || method > 2) return rfail(R, -4, "bad method");
    if (!pose) return rfail(R, -1, "null pose pointer");
    hipSetDevice(R->p.device);
    RigSums S;
    const int rc = rig_eval(R, level, pose, method, &S);
    if (rc) return rc;
    if (err2_split) { err2_split[0] = S.e2p; err2_split[1] = S.e2d; }
    if (n_split) { n_split[0] = S.np; n_split[1] = S.nd; }
    if (H) memcpy(H, S.H, sizeof(S.H));
    if (g) memcpy(g, S.g, sizeof(S.g));
    if (H64) memcpy(H64, S.H64, sizeof(S.H64));
    if (g64) memcpy(g64, S.g64, sizeof(S.g64));
    if (n_rows) *n_rows = S.rows;
    return 0;
}

// RegisterRGBD360.h:383-500 (with fix A).  Returns 0 / RGBD360_ILL_POSED (pose_out = the pose reached, like `rigidTransf = pose_estim;
// return false`).  res->hessian = the last summed Hessian (informationM), res->iters = accepted steps per level, res->err_final =
// the error (sum of squared weighted residuals) at the returned pose.
int rgbd360_rig_align(rgbd360_rig* R, const float guess[16], int method, float pose_out[16], rgbd360_result* res) {
    if (!R) return -1;
    if (!R->have_src || !R->have_trg) return rfail(R, -2, "rgbd360_rig_set_target and _set_source must be called first");
    if (method < 0 || method > 2) return rfail(R, -4, "bad method");
    if (!guess || !pose_out) return rfail(R, -1, "null pose pointer");
    hipSetDevice(R->p.device);
    rgbd360_result Rs;
    memset(&Rs, 0, sizeof(Rs));
    float pose_estim[16], pose_estim_temp[16];
    memcpy(pose_estim, guess, sizeof(pose_estim));
    float Hessian[36] = {0}, Gradient[6] = {0};
    int status = 0, rc = 0;
    double final_error = 0;
    for (int level = R->p.n_pyr - 1; level >= 0 && status == 0; --level) {
        float lambda = 0.001f;                   // RegisterRGBD360.h:389 (double, used as a float scalar by Eigen)
        const double step = 10;
        const unsigned LM_maxIters = 1;
        int it = 0;
        const int maxIters = 10;
        const double tol_residual = pow(10, -1), tol_update = pow(10, -6);
        float update_pose[6] = {1, 1, 1, 1, 1, 1};
        RigSums at_pose, cand;
        if ((rc = rig_eval(R, level, pose_estim, method, &at_pose)) != 0) return rc;      // error; doubles as the first H,g pass
        double error = at_pose.error();
        double diff_error = error;
        auto unorm = [&]() {
            float s2 = 0;
            for (int i = 0; i < 6; ++i) s2 += update_pose[i] * update_pose[i];
            return sqrtf(s2);
        };
        while (it < maxIters && unorm() > tol_update && diff_error > tol_residual) {
            memcpy(Hessian, at_pose.H, sizeof(Hessian));
            memcpy(Gradient, at_pose.g, sizeof(Gradient));
            float M[36];
            for (int k = 0; k < 36; ++k) M[k] = Hessian[k];
            for (int i = 0; i < 6; ++i) M[i * 6 + i] = Hessian[i * 6 + i] + lambda * Hessian[i * 6 + i];
            if (gn::rank6(M) != 6 || !rig_lm_update(Hessian, Gradient, lambda, pose_estim, pose_estim_temp, update_pose)) {
                status = 1;                      // "The problem is ILL-POSED"   RegisterRGBD360.h:443-449
                break;
            }
            if ((rc = rig_eval(R, level, pose_estim_temp, method, &cand)) != 0) return rc;      // FIX A: at pose_estim_temp
            double new_error = cand.error();
            diff_error = error - new_error;
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
                    if (!rig_lm_update(Hessian, Gradient, lambda, pose_estim, pose_estim_temp, update_pose)) break;
                    if ((rc = rig_eval(R, level, pose_estim_temp, method, &cand)) != 0) return rc;
                    new_error = cand.error();
                    diff_error = error - new_error;
                    if (diff_error > 0) {
                        memcpy(pose_estim, pose_estim_temp, sizeof(pose_estim));
                        error = new_error;
                        it = it + 1;
                        at_pose = cand;
                    }
                    LM_it = LM_it + 1;
                }
            }
        }
        Rs.iters[level & 7] = it;
        final_error = error;
    }
    memcpy(pose_out, pose_estim, sizeof(pose_estim));
    Rs.status = status;
    Rs.err_final = final_error;
    memcpy(Rs.hessian, Hessian, sizeof(Hessian));
    memcpy(Rs.gradient, Gradient, sizeof(Gradient));
    if (res) *res = Rs;
    return status;
}

}  // extern "C"
