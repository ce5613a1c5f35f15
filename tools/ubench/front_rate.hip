// Vector-issue cost of the per-pixel pass's two halves on register-resident data (no memory traffic): how many nanoseconds of a SIMD does
// one wave-step of the warp front end (warp_pixel_rc) / of the arithmetic half (consume_stage) take with 1, 2 and 4 waves per SIMD?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -I rgbd360_amd/csrc -o front_rate tools/ubench/front_rate.hip
#include "photo_icp_kernels.h"
#include <cstdio>
#include <vector>
using namespace r360;
#ifndef FRONT_VARIANT
#define FRONT_VARIANT 0
#endif
template <int WHAT>   // 0 front only, 1 consume only, 2 both
__global__ __launch_bounds__(1024) void k(LevelDev lv, EvalConsts ec, Pose16 pose, float* __restrict__ out, int iters) {
    const PoseRT T = load_pose(pose.v);
    const WarpConsts wc = make_warp_consts(T, lv);
    const int tid = threadIdx.x;
    float px = 0.001f * tid, py = 1.f + 0.002f * tid, pz = 2.f - 0.001f * tid;
    EvalAcc A;
    for (int k2 = 0; k2 < 27; ++k2) A.acc[k2] = 0.f;
    A.e2p = A.e2d = 0.f; A.nP = A.nD = A.nVis = 0;
    float sink = 0.f;
    for (int it = 0; it < iters; ++it) {
        PixW w;
        int tr = 0, tc = 0;
        if (WHAT != 1) {
            warp_pixel_rc(T, wc, px, py, pz, lv, w.X, w.Y, w.Z, w.rho2, w.d2, tr, tc, w.vis, w.inv_rho);
            sink += (float)(tr + tc);
        } else {
            w.X = px; w.Y = py; w.Z = pz; w.rho2 = py * py + pz * pz; w.d2 = w.rho2 + px * px; w.inv_rho = __builtin_amdgcn_rsqf(w.rho2); w.vis = ~0ull;
        }
        if (WHAT != 0) {
            w.isrc = 0.3f + 1e-4f * it;
            w.tp.a = 0.31f + px * 1e-3f; w.tp.b = 0.2f + py * 1e-3f; w.tp.c = -0.3f + pz * 1e-3f;
            w.td = w.tp;
            consume_stage<0, true>(w, lv, ec, A);
        }
        px += 1e-3f; py -= 1e-3f; pz += 2e-3f;
        asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));
    }
    float s = sink + A.e2p;
    for (int k2 = 0; k2 < 27; ++k2) s += A.acc[k2];
    out[blockIdx.x * blockDim.x + tid] = s + (float)(A.nP + A.nVis);
}
int main() {
    LevelDev lv; lv.rows = 1024; lv.cols = 2048; lv.n = 2048 * 1024; lv.half_nRows = 511.5f; lv.angle_res_inv = 2048.f / 6.2831853f; lv.pi_k = 1024.f;
    lv.src = nullptr; lv.trgP = nullptr; lv.trgD = nullptr;
    EvalConsts ec; ec.sigma_photo = 0.02f; ec.sigma_depth = 0.05f; ec.thr_photo = 0.01f; ec.thr_depth = 0.01f; ec.sigma_photo_inv_f = 50.f; ec.sigma_photo_inv_d = 50.0;
    Pose16 P; for (int i = 0; i < 16; ++i) P.v[i] = (i % 5 == 0) ? 1.f : 0.01f * i;
    float* out; hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int what = 0; what < 3; ++what)
        for (int threads : {256, 512, 1024}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (what == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, lv, ec, P, out, iters);
                else if (what == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, lv, ec, P, out, iters);
                else hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, lv, ec, P, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            }
            const int wps = threads / 256;
            printf("%s waves/SIMD %d: %.3f ms -> %.1f ns of a SIMD per wave-step\n", what == 0 ? "front  " : what == 1 ? "consume" : "both   ", wps, best,
                   best * 1e6 / (iters * (double)wps));
        }
    return 0;
}
