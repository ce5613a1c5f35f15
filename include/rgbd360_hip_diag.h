/* rgbd360_hip_diag.h -- measurement and self-test entry points of librgbd360_hip.so (RGBD360_DIAG).
 *
 * Not part of the drop-in boundary (include/rgbd360_hip.h): nothing the reference's RegisterPhotoICP / Frame360 callers would
 * bind.  bench.py uses the forced schedule and the kernel timers, the GPU test-suite the arithmetic self-test; a deployment
 * may strip them.
 */
#ifndef RGBD360_HIP_DIAG_H
#define RGBD360_HIP_DIAG_H

#include "rgbd360_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Forced schedule for throughput measurement: n_iters Gauss-Newton iterations on `level`
 * starting at pose0, every step applied regardless of the accept rule, no host round trip.  One iteration =
 * one launch of k_eval_fs (the solve of the previous pass + the fused pass; the last solve in a one-block launch of its own), or one
 * fused pass + one solve launch under rgbd360_debug_set_schedule(ctx, 0, 1).  Enqueued on the context's stream; *elapsed_ms (may be NULL) is the HIP
 * event time around the n_iters iterations (NULL: no events are recorded and the call waits for its result the way
 * rgbd360_align360 does, csrc/host_wait.h). */
int rgbd360_forced_iters(rgbd360_ctx* ctx, int level, const float pose0[16], int method, int n_iters,
                         float pose_out[16], double* last_rms, float* elapsed_ms);
/* The same forced schedule in the lock-step sequence engine's regime (csrc/sequence_engine.h): n_pairs (<= 32) copies of ONE pair
 * (host images as in rgbd360_set_target / _set_source, both frames with the same strides) iterate side by side, every {pass, solve}
 * launch serving all of them.  *elapsed_ms = HIP event time around the n_iters iterations of all pairs; poses_out (may be NULL):
 * n_pairs x 16 floats, the pose every pair reached (identical for all of them, and identical to rgbd360_forced_iters').
 * pass_avg_us (may be NULL): average duration of ten back-to-back launches of the batch pass alone (all n_pairs slots) afterwards. */
int rgbd360_forced_iters_batch(rgbd360_ctx* ctx, int n_pairs, const uint8_t* rgb_trg, const void* depth_trg, const uint8_t* rgb_src,
                               const void* depth_src, size_t rgb_step, size_t depth_step, int depth_type, int rows, int cols,
                               int level, const float pose0[16], int method, int n_iters, float* poses_out, float* elapsed_ms,
                               float* pass_avg_us);
/* One solve on a hand-made partial table (row 0 = `row`: 21 upper-triangle terms of H, 6 of g, the two error sums, the three
 * counts; every other row zero) at the identity pose on `level`, through the two-launch form (fused = 0: k_solve) or the fused form
 * (fused = 1: the prologue of k_eval_fs; the state is read as that launch leaves it; fused = 2: the same with the row at table row 40
 * and a launch that was told to expect one pending row -- the device checks the host's bound and fetches the rest, same result; needs
 * a level of more than 40 block rows).  Drives the state-machine paths real images hardly ever reach (ILL-POSED).
 * out_i: {status, done, level_active, it, n_evals, pend_nb}; cand_out / update_out (may be NULL): the state's candidate pose / update. */
int rgbd360_debug_solve_partials(rgbd360_ctx* ctx, int level, const double row[32], int method, int fused, int out_i[6],
                                 float cand_out[16], float update_out[6]);
/* Average duration in microseconds of `reps` back-to-back launches of the fused per-pixel kernel alone
 * (HIP events on the stream the kernel is launched on). want_hg = 0 times the error-only variant, want_hg = 2 the launch of the
 * single-pair product schedule: k_eval_fs in the forced schedule, i.e. {solve of the previous pass, pass} = one whole Gauss-Newton
 * iteration per launch. */
int rgbd360_time_eval_kernel(rgbd360_ctx* ctx, int level, const float pose[16], int method, int want_hg, int reps,
                             float* avg_us);

/* The same timer with the launches rotating over n_ctx contexts of one device (each with its own copy of a frame pair) on
 * ctxs[0]'s stream: once n_ctx x the level's working set exceeds the 256 MiB Infinity Cache every launch is fed from HBM. */
int rgbd360_time_eval_kernel_rotating(rgbd360_ctx* const* ctxs, int n_ctx, int level, const float pose[16], int method,
                                      int want_hg, int reps, float* avg_us);

/* Same for the solve launch (mode 0: reduction + Gauss-Newton step, forced; mode 1: reduction only), re-using the
 * partials of the last pass. */
int rgbd360_time_solve_kernel(rgbd360_ctx* ctx, int level, int mode, int reps, float* avg_us);

/* Device self-test of the correctly rounded sqrt / reciprocal sequences the warp front end uses: compares them
 * with the compiler's IEEE sqrtf and 1.f/x for the `count` float bit patterns starting at `first_bits`;
 * mismatches[0] = sqrt, mismatches[1] = reciprocal, mismatches[2] = the round-half-up float->int conversion
 * against floor((double)x + 0.5) for |x| < 1e9 (both signs). */
/* Test hooks: the alternative schedules the parity tests hold against the default ones (poses, iteration counts and status must be
 * bit-identical).  fused_solve 0: every Gauss-Newton iteration as a {k_eval, k_solve} launch pair instead of one k_eval_fs launch;
 * fused_occ 0: the occlusion-aware iterations as {k_occ_build, k_eval_occ, k_solve} triples.  Not while an alignment is in flight. */
int rgbd360_debug_set_schedule(rgbd360_ctx* ctx, int fused_solve, int fused_occ);
/* route 1: rgbd360_align360_batch runs every sequence over the per-context route (one context per sub-chunk of pairs; what the
 * occlusion-aware sequences always use) with at most max_contexts contexts (0: the default cap); route 0: the lock-step engines. */
int rgbd360_debug_set_sequence_route(rgbd360_ctx* ctx, int route, int max_contexts);
/* 1 when the library was built with -DRGBD360_DEBUG_KNOBS (csrc/knobs.h: it then reads the A/B environment variables of the measurement tools) */
int rgbd360_debug_knobs_enabled(void);

/* Stage times of rgbd360_frame_planes[_dev], measured with HIP events ON THE CONTEXT'S STREAM at the stage
 * boundaries: us[0] the kernel that forms the organised cloud (and the depth-change mask) = a13, us[1] distance map + normal map = a14,
 * us[2] plane stage (link flags ... hull records, colour when an image is registered) = a15.  _stage_timing(ctx, 1) arms the context's
 * later calls (four event records per call), _stage_times reads the last call's.  With the refinement on, a15 ends at the last kernel in
 * front of the refinement (the refinement synchronises with the host).  bench.py's `roofline_frame360` block. */
int rgbd360_frame_planes_stage_timing(rgbd360_ctx* ctx, int on);
int rgbd360_frame_planes_stage_times(rgbd360_ctx* ctx, float us[3]);

int rgbd360_selftest_math(rgbd360_ctx* ctx, uint32_t first_bits, uint32_t count, unsigned long long mismatches[3]);
/* csrc/libm_f32.h (asinf / atanf / roundf / atan2f restated operation for operation, what rgbd360_set_index_arithmetic(ctx, 1) computes
 * with) as the DEVICE evaluates it, against the C library of this process: the floats first_bits .. first_bits + count - 1 through the
 * one-argument functions (asinf where |x| <= 1.5), `count` drawn pairs through atan2f.  mismatches[4] = {asinf, atanf, roundf, atan2f}. */
int rgbd360_selftest_libm(rgbd360_ctx* ctx, uint32_t first_bits, uint32_t count, unsigned long long mismatches[4]);

#ifdef __cplusplus
}
#endif
#endif /* RGBD360_HIP_DIAG_H */
