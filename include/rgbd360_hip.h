/*
 * rgbd360_hip.h -- C ABI of the MI355X (gfx950) dense spherical RGB-D alignment library.
 *
 * Drop-in boundary for ONE path of EduFdez/rgbd360: RegisterPhotoICP's spherical alignment
 * (setTargetFrame / setSourceFrame / alignFrames360 with its three occlusion modes and the per-pixel passes they
 * call), the pinhole single-sensor alignFrames, and the adjacent Frame360 per-pixel stages (frame file reader,
 * 8-sensor spherical stitching, sphere cloud, normal map, planar regions).  Everything is plain C: opaque context, POD structs, raw
 * pointers and sizes.  File:line citations are relative to the reference tree; "RPI.h" is
 * include/RegisterPhotoICP.h.
 *
 * Conventions
 *   - 4x4 poses and the 6x6 Hessian are COLUMN-MAJOR float arrays (Eigen's default layout, which
 *     is what Eigen::Matrix4f::data() of the reference's relPose/hessian hands over).
 *   - relPose maps source-frame points into the target frame: p_trg = R p_src + t (RPI.h:2663).
 *   - Images are row-major with an explicit byte stride (cv::Mat::step).  rgb is 8UC3, depth is
 *     16UC1 millimetres (depth_type 0; Frame360.h:394) or 32FC1 metres (depth_type 1; RPI.h:318).
 *   - Host image pointers are copied before the call returns; the caller keeps ownership
 *     (the reference aliases cv::Mat buffers, RPI.h:296,319, and never frees caller memory).
 *   - A context is stateful and not re-entrant, like a RegisterPhotoICP object (one per thread).
 *   - Every function returns 0 on success, a positive rgbd360 status (below) for algorithmic
 *     outcomes and a negative value for HIP/argument errors; rgbd360_last_error() has the text.
 *   - There is NO CPU fallback: without a usable HIP device rgbd360_create fails.
 */
#ifndef RGBD360_HIP_H
#define RGBD360_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rgbd360_ctx rgbd360_ctx;

/* costFuncType, RPI.h:194 */
enum { RGBD360_PHOTO_CONSISTENCY = 0, RGBD360_DEPTH_CONSISTENCY = 1, RGBD360_PHOTO_DEPTH = 2 };

/* status codes */
enum {
    RGBD360_OK = 0,
    RGBD360_ILL_POSED = 1,      /* rank(H + lambda diag H) != 6, RPI.h:4682-4690: pose_out = last accepted pose */
    RGBD360_NO_VALID_PIXELS = 2 /* the error pass found no residual (the reference would divide by zero) */
};

/* Replaces the constructor defaults + setters of RegisterPhotoICP (RPI.h:201-221, 224-269) and the
 * loop constants of alignFrames360 (RPI.h:4593-4595). */
typedef struct {
    int   n_pyr;            /* setNumPyr, default 4 */
    float min_depth;        /* setMinDepth, 0.3 m */
    float max_depth;        /* setMaxDepth, 6.0 m */
    float sigma_photo;      /* setGrayVariance (sets the std-dev, RPI.h:242-245), 6/255 */
    float sigma_depth;      /* setDepthVariance (std-dev, RPI.h:248-251), 0.2 */
    float thres_sal_photo;  /* thresSaliencyIntensity, 0.01 */
    float thres_sal_depth;  /* thresSaliencyDepth, 0.01 */
    int   max_iters;        /* 10 */
    float tol_residual;     /* 1e-3 */
    float tol_update;       /* 1e-4 */
    int   mask_seams;       /* 1: zero the gradient bands at the 8-sensor seams (RPI.h:4538-4549) */
    int   device;           /* HIP device ordinal */
} rgbd360_params;

/* What callers read from a RegisterPhotoICP after alignFrames360: getHessian()/getGradient()
 * (RPI.h:279-288), SSO, avPhotoResidual/avDepthResidual (RPI.h:180-189), num_iterations (RPI.h:177). */
typedef struct {
    int    status;
    int    iters[8];        /* accepted Gauss-Newton iterations per pyramid level (index = level) */
    float  sso;             /* visible pixels / image size at level 0 (RPI.h:3226) */
    double err_final;       /* RMS residual of the error pass at pose_out, level 0 (occlusion 1 / 2 and the pinhole path:
                               avPhotoResidual + avDepthResidual, as their error functions define it) */
    double rms_photo;       /* photo / depth RMS of that pass (the reference leaves these unset on this path) */
    double rms_depth;
    float  hessian[36];     /* column-major 6x6: H of the last calcHessGrad_sphere the reference would have run */
    float  gradient[6];
} rgbd360_result;

void rgbd360_default_params(rgbd360_params* p);

/* RegisterPhotoICP::RegisterPhotoICP() (RPI.h:201) */
int  rgbd360_create(const rgbd360_params* p, rgbd360_ctx** out);
void rgbd360_destroy(rgbd360_ctx* ctx);
const char* rgbd360_last_error(rgbd360_ctx* ctx);

/* RegisterPhotoICP::setTargetFrame(cv::Mat& rgb, cv::Mat& depth) (RPI.h:498-516): gray conversion,
 * gray + depth pyramids, gradient pyramids.  Host pointers. */
int rgbd360_set_target(rgbd360_ctx* ctx, const uint8_t* rgb, size_t rgb_step, const void* depth, size_t depth_step,
                       int depth_type, int rows, int cols);
/* RegisterPhotoICP::setSourceFrame (RPI.h:480-494) + the per-level LUT of 3-D points (RPI.h:4554-4587). */
int rgbd360_set_source(rgbd360_ctx* ctx, const uint8_t* rgb, size_t rgb_step, const void* depth, size_t depth_step,
                       int depth_type, int rows, int cols);
/* Same, images already resident in device memory (HBM) on the context's device. */
int rgbd360_set_target_dev(rgbd360_ctx* ctx, const uint8_t* rgb_dev, size_t rgb_step, const void* depth_dev,
                           size_t depth_step, int depth_type, int rows, int cols);
int rgbd360_set_source_dev(rgbd360_ctx* ctx, const uint8_t* rgb_dev, size_t rgb_step, const void* depth_dev,
                           size_t depth_step, int depth_type, int rows, int cols);
/* Odometry reuse: the previous source frame becomes the target without re-uploading
 * (OdometryRGBD360.cpp:189-190 re-sets both frames every step). Builds the target gradients on device. */
int rgbd360_promote_source_to_target(rgbd360_ctx* ctx);

/* The arithmetic of the warp -- spherical (RPI.h:2663-2684 / 2959-2989: p' = R p + t, phi = asin(x / |p'|), theta = atan2(y, z) + PI,
 * row / column = round(...)) and pinhole (RPI.h:701-708: column = round(x fx / z + ox), ...) -- for every later pass of this context:
 * rgbd360_align360 and its _begin / _finish, the batch and sequence entries that run on this context, the occlusion-aware passes,
 * rgbd360_eval*, rgbd360_warp_indices*, rgbd360_align_pinhole; the sibling contexts and engines a sequence call creates inherit it, a
 * multi-GPU handle has rgbd360_multi_set_index_arithmetic (the 8-sensor rgbd360_rig_* objects keep the device definition):
 *   0 (default): the device definition -- fused multiply-adds, two correctly rounded arctangent evaluations by one polynomial,
 *      round-half-up; mirrored bit for bit by the CPU checker's math_mode 1.  About 1e-4 of the pixels land on a neighbouring target
 *      pixel compared with the reference built against glibc.
 *   1: the REFERENCE's arithmetic -- Eigen's product order without fused multiply-adds, norm(), 1 / dist, asinf, atan2f + (double) PI,
 *      roundf, the two functions restated operation for operation from glibc 2.35's fdlibm float code (csrc/libm_f32.h; neither is
 *      correctly rounded, so "the same index" means that operation sequence).  Target indices, visibility and |p'|^2 are bit-equal to
 *      the reference's own; the spherical per-pixel pass costs ~1.8 x.  The pinhole warp has no transcendental: there the option means
 *      the reference's operation order, 1.0 / z in double and roundf.
 * Returns -6 while an alignment is in flight. */
int rgbd360_set_index_arithmetic(rgbd360_ctx* ctx, int mode);
int rgbd360_get_index_arithmetic(rgbd360_ctx* ctx);

/* RegisterPhotoICP::alignFrames360(pose_guess, method, occlusion) (RPI.h:4519-4784) + getOptimalPose().
 * occlusion 0 = regular registration, 1 / 2 = the z-buffer variants errorPhotoICP_sphereOcc1/2 + calcHessGrad_sphereOcc1/2
 * (RPI.h:3232-4249) with the sequential semantics of the reference source (its OpenMP build races on the z-buffer; see
 * DESIGN.md).  occlusion 1 needs method PHOTO_DEPTH: with one modality the reference's error is 0/0 and it returns the
 * guess (status RGBD360_NO_VALID_PIXELS here). */
int rgbd360_align360(rgbd360_ctx* ctx, const float guess[16], int method, int occlusion, float pose_out[16],
                     rgbd360_result* res);

/* The same alignment split in two, so that several contexts (one per frame pair, each on its own HIP stream) can be in
 * flight on one GPU at once: _begin enqueues the coarse-to-fine schedule and returns without waiting; _finish waits,
 * tops the schedule up if a level needed more iterations than were enqueued, and returns what rgbd360_align360 returns.
 * Coarse-level launches of different pairs then overlap on the device (they fill only a fraction of the CUs). */
int rgbd360_align360_begin(rgbd360_ctx* ctx, const float guess[16], int method, int occlusion);
int rgbd360_align360_finish(rgbd360_ctx* ctx, float pose_out[16], rgbd360_result* res);

/* A sequence of n_frames frames = n_frames-1 consecutive pairs (pair j: frame j = target, frame j+1 = source), the way
 * OdometryRGBD360.cpp:141-297 walks a sequence; SURVEY.md 8b/8e's batch entry for ONE GPU (several GPUs: rgbd360_multi_* below,
 * or one process per GPU each calling this on its contiguous shard, rgbd360_amd/batch.py).  rgb[k] / depth[k]: host images as
 * in rgbd360_set_target.  n_inflight (1..64) pairs are in flight: the sequence is cut into that many contiguous spans ("slots")
 * which advance in lock step -- every kernel launch of a round (frame set-up, each pass and solve of each pyramid level)
 * carries a slot dimension, so a round of n_inflight alignments costs the launch count of one (csrc/sequence_engine.h;
 * 32 = two engines of 16 slots is the measured optimum at 2048x1024, ~3.6 GB of HBM; 16 costs 5 %; host frames use at most 16).  Inside a span every frame is uploaded once,
 * one round ahead of its alignment on a copy stream: the host images must stay unchanged until the call returns.  Poses are
 * bit-identical to rgbd360_align360 pair by pair, whatever n_inflight.  The occlusion-aware variants run one context per span
 * instead (at most six, three when GPU_MAX_HW_QUEUES > 4: the measured optima of that route).  guess (NULL = identity) is the initial pose of every pair.  poses_out: (n_frames-1) x 16
 * floats column-major; results_out (may be NULL): n_frames-1 records whose .status carries the per-pair outcome (0 / ILL_POSED
 * / NO_VALID_PIXELS).  Returns 0, or the first negative error. */
int rgbd360_align360_batch(rgbd360_ctx* ctx, int n_frames, const uint8_t* const* rgb, size_t rgb_step, const void* const* depth,
                           size_t depth_step, int depth_type, int rows, int cols, const float guess[16], int method,
                           int occlusion, int n_inflight, float* poses_out, rgbd360_result* results_out);
/* The same with every rgb[k] / depth[k] already in HBM on the context's device (as rgbd360_set_target_dev): no PCIe traffic
 * inside the call; poses_out / results_out stay host arrays. */
int rgbd360_align360_batch_dev(rgbd360_ctx* ctx, int n_frames, const uint8_t* const* rgb, size_t rgb_step, const void* const* depth,
                               size_t depth_step, int depth_type, int rows, int cols, const float guess[16], int method,
                               int occlusion, int n_inflight, float* poses_out, rgbd360_result* results_out);

/* ---- one process, several GPUs (BASELINE.json configs[3]) ---------------------------------------------------
 * The sequence path shards by independent frame pairs: device d gets the contiguous pairs rgbd360_shard_range(n_frames-1, d,
 * n_gpus) and therefore the frames lo..hi (one boundary frame is shared by two neighbours); one host thread per device drives
 * that device's contexts as rgbd360_align360_batch does, no collective touches the data path, and ONE ncclAllGather (RCCL over
 * xGMI) of the per-pair result rows {pose[16], rgbd360_result} ends the call, after which every device holds the whole
 * trajectory (the host reads device 0's copy and checks it against the rows the shards produced).  The caller composes the
 * relative poses like OdometryRGBD360.cpp:257.  n_gpus == 1 runs without any RCCL call (RGBD360_FORCE_RCCL=1 in the
 * environment forces the one-rank exchange, for tests).  The handle owns the per-device contexts and the communicators:
 * create it once, align many sequences.  Replaces the role of the reference's single-threaded odometry loop
 * (Registration/OdometryRGBD360.cpp:141-297) for a recorded sequence. */
typedef struct rgbd360_multi rgbd360_multi;
/* device_ids: n_gpus distinct HIP device indices, NULL = 0..n_gpus-1.  Returns 0, -100 no HIP device, -101 bad device list,
 * -105 RCCL initialisation failed. */
int  rgbd360_multi_create(const rgbd360_params* p, int n_gpus, const int* device_ids, rgbd360_multi** out);
void rgbd360_multi_destroy(rgbd360_multi* m);
const char* rgbd360_multi_last_error(rgbd360_multi* m);
int  rgbd360_multi_n_gpus(rgbd360_multi* m);
int  rgbd360_multi_uses_rccl(rgbd360_multi* m);
/* rgbd360_set_index_arithmetic on every device's context of the handle (0: device definition, 1: the reference's libm arithmetic). */
int  rgbd360_multi_set_index_arithmetic(rgbd360_multi* m, int mode);
/* Contiguous balanced partition used by the sharding: the first n_items % world ranks get one item more. */
void rgbd360_shard_range(int n_items, int rank, int world, int* lo, int* hi);
/* Layout of the exchange step (ncclAllGather wants equal counts): every rank contributes rows_per_rank = ceil(n_pairs / world) result
 * rows, the first hi - lo of them its pairs; global pair `pair` is row *row of the gathered table and belongs to rank *rank.
 * Pure index arithmetic (no device): what rgbd360_multi_* use to scatter the gathered rows back into sequence order. */
void rgbd360_gather_slot(int n_pairs, int world, int pair, int* rank, int* row, int* rows_per_rank);
/* Host frames (as rgbd360_align360_batch: uploaded one frame ahead on each device's copy stream, unchanged until the call
 * returns).  poses_out: (n_frames-1) x 16 floats; results_out may be NULL.  Returns 0 or the first negative error. */
int  rgbd360_multi_align_sequence(rgbd360_multi* m, int n_frames, const uint8_t* const* rgb, size_t rgb_step,
                                  const void* const* depth, size_t depth_step, int depth_type, int rows, int cols,
                                  const float guess[16], int method, int occlusion, int n_inflight, float* poses_out,
                                  rgbd360_result* results_out);
/* Resident variant: _load_sequence copies every device's frames lo..hi into its HBM once (sized for 288 GB per device: a
 * 2048x1024 frame is 10.5 MB); _align_resident then aligns the whole sequence with no PCIe traffic but the result rows. */
int  rgbd360_multi_load_sequence(rgbd360_multi* m, int n_frames, const uint8_t* const* rgb, size_t rgb_step,
                                 const void* const* depth, size_t depth_step, int depth_type, int rows, int cols);
int  rgbd360_multi_align_resident(rgbd360_multi* m, const float guess[16], int method, int occlusion, int n_inflight,
                                  float* poses_out, rgbd360_result* results_out);
/* One-shot form (create + align_sequence + destroy): pays the communicator set-up on every call. */
int  rgbd360_align360_batch_multi(const rgbd360_params* p, int n_frames, const uint8_t* const* rgb, size_t rgb_step,
                                  const void* const* depth, size_t depth_step, int depth_type, int rows, int cols,
                                  const float guess[16], int method, int occlusion, int n_inflight, int n_gpus,
                                  const int* device_ids, float* poses_out, rgbd360_result* results_out);

/* ---- stage-level entry points (parity tests and measurement) ------------------------------------------------ */

/* Pyramid planes as float32 rows x cols (level dims via rgbd360_level_dims).
 * which: 0 graySrc 1 grayTrg 2 depthSrc 3 depthTrg 4 grayTrgGradX 5 grayTrgGradY 6 depthTrgGradX 7 depthTrgGradY
 * (the public pyramid vectors of RPI.h:198-199).  Gradients already carry the seam mask. */
int rgbd360_level_dims(rgbd360_ctx* ctx, int level, int* rows, int* cols);
int rgbd360_get_plane(rgbd360_ctx* ctx, int which, int level, float* host_out);
/* LUT_xyz_sphere of a level (RPI.h:4554-4587) as n x 3 floats; invalid points have x = -10000. */
int rgbd360_get_lut(rgbd360_ctx* ctx, int level, float* host_out_xyz);

/* One fused per-pixel pass at `pose` on `level`: errorPhotoICP_sphere (RPI.h:2545) and calcHessGrad_sphere
 * (RPI.h:2745) evaluated at the same pose.  Any output pointer may be NULL.
 * err2 / n_valid: sum of squared weighted residuals and their count (RPI.h:2707-2729), split photo / depth in
 * err2_split[2], n_split[2].  H (column-major) and g from float64 block partials of float32 rows. */
int rgbd360_eval(rgbd360_ctx* ctx, int level, const float pose[16], int method, double* err2, long long* n_valid,
                 double err2_split[2], long long n_split[2], float H[36], float g[6], double H64[36], double g64[6],
                 long long* n_visible);
/* The same pass in the occlusion-aware variants (occlusion 1: RPI.h:3232-3716, 2: RPI.h:3720-4249; 0 = rgbd360_eval).
 * n_split[] are the reference's nValidPhotoPts / nValidDepthPts (occlusion 2: both = nValidDepthPts), the error of
 * the alignment loop is sqrt(err2_split[0]/n_split[0]) + sqrt(err2_split[1]/n_split[1]). */
int rgbd360_eval_occ(rgbd360_ctx* ctx, int level, const float pose[16], int method, int occlusion, double* err2,
                     long long* n_valid, double err2_split[2], long long n_split[2], float H[36], float g[6], double H64[36],
                     double g64[6], long long* n_visible);
/* Warped target pixel of every source pixel of `level` under `pose`: out[2i] = r', out[2i+1] = c', -1 if the
 * pixel is invalid or leaves the image (RPI.h:2663-2684).  Host output, n x 2 int32. */
int rgbd360_warp_indices(rgbd360_ctx* ctx, int level, const float pose[16], int32_t* host_out_rc);
/* One Gauss-Newton step on the device from the H,g of the preceding rgbd360_eval... exposed for tests:
 * pose_tmp = exp(-H^-1 g) * pose (RPI.h:4682-4697).  Returns RGBD360_ILL_POSED when the rank test fails. */
int rgbd360_gn_step(rgbd360_ctx* ctx, const float H[36], const float g[6], float lambda, const float pose[16],
                    float pose_tmp[16], float update[6]);

/* Measurement and self-test entry points (forced iteration schedule, kernel timers, device arithmetic self-test) are not part
 * of the product ABI: include/rgbd360_hip_diag.h. */

/* The HIP stream all work of this context is enqueued on (hipStream_t as void*). */
void* rgbd360_stream(rgbd360_ctx* ctx);
int   rgbd360_sync(rgbd360_ctx* ctx);
/* Number of HIP devices visible; does not create a context. */
int   rgbd360_device_count(void);

/* ---- dense registration of two frames of the 8-sensor rig ---------------------------------------------
 * RegisterRGBD360::RegisterDensePhotoICP (RegisterRGBD360.h:344-520) over RegisterPhotoICP::calcPhotoICPError_robot
 * (RPI.h:4905-5076) and calcHessianGradient_robot (RPI.h:5083-5407): the unknown is the RIG's relative pose (p_rig1 = T p_rig2),
 * every sensor contributes its pinhole photometric / depth rows through its extrinsic Rt_s (sensor -> rig), the 8 sets of normal
 * equations add up, Levenberg-Marquardt on the sum (lambda 0.001, x / 10, one retry, full SE(3) exponential, tolerances 0.1 on
 * the summed squared error / 1e-6).  The reference function is broken as written; three defects are FIXED here and in the
 * oracle (oracle/photo_icp_ref.cpp documents them): new_error is evaluated at the candidate pose (RegisterRGBD360.h:462,488
 * use the old one), jacobianRt_z is row 2 of the transform Jacobian (RPI.h:5226-5228 leave it uninitialised), and the depth
 * residual uses the transformed point's depth (RPI.h:5037 uses the untransformed one).
 * Rt: n_sensors x 16 floats, column-major sensor -> rig poses (Calib360::Rt_); fx, fy, ox, oy: the sensors' level-0 intrinsics
 * (RegisterRGBD360.h:357-365: 525 * width / 640, centre).  All sensor images of a frame share one size. */
typedef struct rgbd360_rig rgbd360_rig;
int  rgbd360_rig_create(const rgbd360_params* p, int n_sensors, const float* Rt, float fx, float fy, float ox, float oy,
                        rgbd360_rig** out);
void rgbd360_rig_destroy(rgbd360_rig* rig);
const char* rgbd360_rig_last_error(rgbd360_rig* rig);
/* frame1 (target) / frame2 (source): n_sensors host images each, as rgbd360_set_target takes them; copied before return. */
int  rgbd360_rig_set_target(rgbd360_rig* rig, const uint8_t* const* rgb, size_t rgb_step, const void* const* depth,
                            size_t depth_step, int depth_type, int rows, int cols);
int  rgbd360_rig_set_source(rgbd360_rig* rig, const uint8_t* const* rgb, size_t rgb_step, const void* const* depth,
                            size_t depth_step, int depth_type, int rows, int cols);
/* One fused pass over all sensors at rig pose `pose`: err2_split = {photo, depth} sums of squared weighted residuals (no
 * saliency test; their sum is calcPhotoICPError_robot summed over the sensors), n_split the pixel counts, H / g the summed
 * normal equations (float, per-sensor totals added in sensor order), H64 / g64 the same in double, n_rows the Jacobian rows. */
int  rgbd360_rig_eval(rgbd360_rig* rig, int level, const float pose[16], int method, double err2_split[2], long long n_split[2],
                      float H[36], float g[6], double H64[36], double g64[6], long long* n_rows);
/* The registration.  Returns 0 or RGBD360_ILL_POSED (pose_out = the pose reached).  res->iters = accepted steps per level,
 * res->hessian = the summed Hessian of the last step (informationM, RegisterRGBD360.h:511), res->err_final = the summed
 * squared error at pose_out. */
int  rgbd360_rig_align(rgbd360_rig* rig, const float guess[16], int method, float pose_out[16], rgbd360_result* res);
/* useSaliency(true) on the per-sensor RegisterPhotoICP objects (RPI.h:266-269): calcPhotoICPError_robot and
 * calcHessianGradient_robot run over vSalientPixels only (RPI.h:4930-5003, 5121-5262: the same loop bodies over the list built by
 * calcGradientXY_saliency from the TARGET's gray gradients, RPI.h:401-425, used as source pixel indices).  Off by default, as in
 * every application of the reference. */
int  rgbd360_rig_use_saliency(rgbd360_rig* rig, int on, float thres_saliency);

/* ---- Frame360 per-pixel stages ------------------------------------------------------------------------------ */

/* Spherical point cloud from a range panorama.
 * convention 0: Frame360::buildSphereCloud_fromImage (Frame360.h:555-612): depth u16 mm, phi offset 31.5 deg,
 *               xyz = d (sin phi, -cos phi sin theta, -cos phi cos theta), NaN where depth == 0.
 * convention 1: Frame360_stereo::buildSphereCloud (Frame360_stereo.h:454-512): depth f32 m valid in (0,15),
 *               phi = (row+166) step - pi/2, theta = col step - pi, xyz = d (sin theta cos phi, sin phi, cos theta cos phi).
 * convention 2: full-sphere RegisterPhotoICP convention (RPI.h:4567-4582) with depth_type as in set_source.
 * Output: xyz as rows*cols x 3 float32 (host).  */
int rgbd360_sphere_cloud(rgbd360_ctx* ctx, const void* depth, size_t depth_step, int depth_type, int rows, int cols,
                         int convention, float* host_out_xyz);

/* Normal map of an organised cloud (rows*cols x 3 float32, NaN = invalid): pcl::IntegralImageNormalEstimation with
 * AVERAGE_3D_GRADIENT, setMaxDepthChangeFactor(max_depth_change_factor), setNormalSmoothingSize(normal_smoothing_size),
 * setDepthDependentSmoothing(true), as configured at Frame360.h:949-957 (0.02, 8) and Frame360_stereo.h:854-862
 * (0.05, 8).  depth_mode 0 uses the z coordinate as "depth" like PCL, 1 the range |p| (full spheres).
 * normals_out: rows*cols x 3, NaN where PCL leaves the normal undefined; normals point towards the origin. */
int rgbd360_normals(rgbd360_ctx* ctx, const float* xyz, int rows, int cols, float max_depth_change_factor,
                    float normal_smoothing_size, int depth_mode, float* normals_out);
/* The chamfer (1 / 1.4) distance-to-depth-discontinuity map that drives the smoothing window (diagnostics). */
int rgbd360_distance_map(rgbd360_ctx* ctx, const float* xyz, int rows, int cols, float max_depth_change_factor,
                         int depth_mode, float* dist_out);

/* pcl::FastBilateralFilter<PointXYZRGBA> with setSigmaS(sigma_s) / setSigmaR(sigma_r) on an organised cloud (rows*cols x 3
 * float32, NaN = invalid): the smoothing Frame360 applies to every sensor cloud before its planes are segmented
 * (Frame360.h:40, 493-499: sigma_s 10 px, sigma_r 0.05 m).  PCL's bilateral-grid algorithm (third-party; restated, see
 * oracle/frame360_ref.cpp): only z changes, x and y are copied.  xyz_out may alias xyz. */
int rgbd360_bilateral_filter(rgbd360_ctx* ctx, const float* xyz, int rows, int cols, float sigma_s, float sigma_r, float* xyz_out);

#define RGBD360_HULL_MAX 64
/* One planar region: n . x + d = 0 with n towards the origin, curvature = lambda_min / trace(cov). */
typedef struct {
    float centroid[3];
    float normal[3];
    float d;
    float curvature;
    int   count;        /* inliers */
    int   root;         /* smallest pixel index of the region = its label */
    /* Extent descriptors -- the roles of mrpt::pbmap::Plane::areaHull / elongation / v3PpalDir (Frame360.h:1025-1037; MRPT is
     * third-party and not in the reference tree):
     *   area        area of the convex hull of the region's contour projected onto its plane (calcConvexHull +
     *               computeMassCenterAndArea): metric, whatever the pixel density.  The device reduces the boundary pixels to the
     *               region's extreme point in each of 1024 in-plane directions (eight interleaved sets of 128), the host runs hull + shoelace on those (an inscribed
     *               polygon: exact for sharp-cornered polygons, 0.01 % low for a disc, a few 0.1 % low where long edges are slightly bowed).  Frame360.h:1031 compares it with min_area_plane (0.12 m2),
     *               RegisterRGBD360.h:126-136 ranks planes by it.
     *   elongation  sqrt(l2 / l1), ppal_dir = eigenvector of l2: PCA of the inliers (l1 <= l2 the in-plane eigenvalues of their
     *               covariance beside the normal's l0), as calcElongationAndPpalDir does on the inlier cloud. */
    float area;
    float elongation;
    float ppal_dir[3];
    /* area_moment  12 sqrt(l1 l2): the rectangle with the inliers' second moments (rounds 1-2 reported this as `area`; pixel-density
     *              weighted, it reads a wall seen from a spherical image several times too small).  rgbd360_merge_planes rebuilds a
     *              piece's covariance from it; 0 in a caller-made record means "take area".
     * center_hull  mass centre of the hull polygon (what computeMassCenterAndArea leaves in v3center); hull_points = vertices of the
     *              hull (0: no hull was formed -- caller-made record, or fewer than three extreme points: area = area_moment then). */
    float area_moment;
    float center_hull[3];
    int   hull_points;
    /* Colour descriptors -- the roles of mrpt::pbmap::Plane::v3colorNrgb / dominantIntensity (calcMainColor2) and hist_H
     * (calcPlaneHistH), which Frame360.h:1045-1046 / Frame360_stereo.h:949-950 fill for every plane and the PbMap matcher's unary
     * colour constraint reads (configLocaliser_spherical.ini:19-21).  Filled when a colour image is registered for the plane call
     * (rgbd360_set_plane_color_image); color_count = 0 otherwise, and the matcher then skips its colour tests.
     *   color_nrgb   mean of the normalised colour (R, G, B) / (R + G + B) over the inliers with R + G + B > 0 (invariant to a global
     *                brightness change), color_dev its standard deviation per channel.  MRPT's calcMainColor2 takes the mean-shift
     *                mode of these values; the mean is what its calcMainColor takes, and the two coincide for a region of one colour.
     *   intensity    mean R + G + B (0..765) of the same pixels.
     *   hist_h       normalised histogram over ALL inliers: 72 bins of 5 degrees of hue for saturated pixels, [72] dark pixels
     *                (V <= 0.2), [73] unsaturated ones (S <= 0.2). */
    int   color_count;
    float color_nrgb[3];
    float color_dev[3];
    float intensity;
    float hist_h[74];
    /* The hull POLYGON itself -- mrpt::pbmap::Plane::polygonContourPtr, which Frame360::mergePlanes tests for proximity vertex by vertex and
     * edge by edge (Frame360.h:680-711) and mergePlane2 re-hulls: hull_n <= RGBD360_HULL_MAX vertices in the frame of the plane record,
     * counter-clockwise seen from the side the normal points to (the camera's side), on the fitted plane.  A hull of more vertices is
     * thinned to its extreme points in RGBD360_HULL_MAX evenly spaced in-plane directions (an inscribed polygon: < 0.2 % of the area of a
     * disc is lost).  hull_n = 0: the record carries no polygon (caller-made record, or no hull was formed). */
    int   hull_n;
    float hull[RGBD360_HULL_MAX][3];
    /* The DOMINANT colour -- what mrpt::pbmap::Plane::calcMainColor2 leaves in v3colorNrgb / dominantIntensity (Frame360.h:1046): the plane's
     * pixels thinned to about 2000 samples, then getMultiDimMeanShift_color: samples farther from the running mean than the norm of the
     * standard deviation are dropped for good, until half the samples are gone or the mean stops moving (third-party; restated in integer
     * arithmetic, so the CPU checker repeats it exactly).  On a surface of two colours (a poster on a wall) the mean lies between them and
     * moves with the share of each in view; the dominant colour does not.  color_mode_count = samples it was sought over (0: none --
     * no colour image, or a caller-made record: the matcher then compares color_nrgb / intensity), color_concentration = the share of the
     * samples it ended on (MRPT's `concentration`). */
    int   color_mode_count;
    float color_mode[3];
    float intensity_mode;
    float color_concentration;
} rgbd360_plane;

/* Registers the colour image that goes with the organised cloud of the context's next plane calls (rgbd360_plane_fit,
 * _frame_planes[_dev], _cloud_planes, _sensor_planes): rgb is rows x cols x 3 uint8 with rgb_step bytes per row, and cloud pixel
 * (r, c) takes the colour of image pixel (r * step + step / 2, c * step + step / 2) -- step 1 for a panorama or a full-resolution
 * sensor cloud, the down-sampling step for a down-sampled one (DownsampleRGBD.h:240, 285-287: the colour of the block's centre
 * pixel).  on_device = 0: rgb is host memory and is copied at once; 1: device memory that must stay valid through those calls.
 * The planes of a call whose cloud does not have (rows / step) x (cols / step) points come back without colour.  rgb = NULL clears.
 * Channel order: the three bytes of a pixel are taken as R, G, B.  The descriptors of two planes are only ever compared channel by
 * channel, so an image in OpenCV's B, G, R order (the reference's panorama, Frame360.h:594-596) gives the same matches -- with
 * color_nrgb / color_mode channel-swapped and hist_h a mirrored hue circle relative to MRPT's; swap the channels first if the records
 * are to be exchanged with an MRPT PbMap. */
int rgbd360_set_plane_color_image(rgbd360_ctx* ctx, const uint8_t* rgb, size_t rgb_step, int rows, int cols, int step, int on_device);

/* Planar regions of an organised cloud with normals: pcl::OrganizedMultiPlaneSegmentation::segment as configured at
 * Frame360.h:958-977 (min_inliers 80, angular 0.0398 rad, distance 0.02) / Frame360_stereo.h:863-882 (40, 0.05, 0.05):
 * PlaneCoefficientComparator (depth-dependent distance threshold) + organised connected components + per-region
 * centroid / covariance / smallest eigenvector / curvature (the values Frame360.h:984-996 copies into
 * mrpt::pbmap::Plane).  labels_out (may be NULL): per pixel the region's root pixel index, -1 for non-finite points.
 * Planes are returned in PCL's order (by first pixel), at most max_planes; when more regions pass the filters the
 * max_planes LARGEST (inlier count) are kept, and rgbd360_planes_available reports how many there were.
 * The inlier sums are exact 64-bit integer sums of terms rounded to 2^-28 m (m^2) -- order independent, ~20 x finer than the float
 * accumulators of PCL 1.7's computeMeanAndCovarianceMatrix; a region a few millimetres across whose smallest eigenvalue lies within
 * ~1e-8 m^2 of max_curvature x trace may still fall on the other side of the filter than a float64 evaluation puts it.  Errors of the
 * plane calls: -7 more than 4096 regions exceed min_inliers; -8 the sums left their range (N r^2 >= 3.4e10 m^2: a whole 4096 x 2048
 * frame that is one region beyond 64 m). */
int rgbd360_plane_fit(rgbd360_ctx* ctx, const float* xyz, const float* normals, int rows, int cols, int min_inliers,
                      float angular_threshold, float distance_threshold, float max_curvature, int depth_mode,
                      int32_t* labels_out, rgbd360_plane* planes_out, int max_planes, int* n_planes_out);

/* Number of regions that passed every filter in the context's last plane call (rgbd360_plane_fit, _frame_planes[_dev],
 * _cloud_planes, _sensor_planes); larger than the returned n_planes when the caller's max_planes cut the list -- grow the
 * buffer and call again. */
int rgbd360_planes_available(rgbd360_ctx* ctx);
/* The `refine` half of pcl::OrganizedMultiPlaneSegmentation::segmentAndRefine (what Frame360.h:977 / :868 / Frame360_stereo.h:882
 * call) for every later plane call of this context: after `segment`, planes grow into neighbouring pixels of regions that did
 * not become planes when the pixel's point lies within distance_threshold of the plane (PCL's refinement comparator: 0.02 m, not
 * depth dependent) -- PCL's two raster passes, solved on the device by Jacobi sweeps to the same labels.  labels_out then holds
 * the refined labels; a plane keeps centroid / normal / d / curvature of `segment` (as PCL's PlanarRegion does), count and the
 * extent descriptors (area, elongation, ppal_dir) follow the grown inlier set.  Off by default (enabled = 0: plain `segment`). */
int rgbd360_set_plane_refinement(rgbd360_ctx* ctx, int enabled, float distance_threshold);
/* Pixels relabelled and Jacobi sweeps of the last plane call with refinement on. */
int rgbd360_plane_refinement_stats(rgbd360_ctx* ctx, int* pixels_relabelled, int* sweeps);

/* Range panorama -> sphere cloud -> normals -> planar regions in one call (cloud and normals stay on the device
 * between the stages); xyz_out / normals_out / labels_out may be NULL. */
int rgbd360_frame_planes(rgbd360_ctx* ctx, const void* depth, size_t depth_step, int depth_type, int rows, int cols,
                         int convention, float max_depth_change_factor, float normal_smoothing_size, int min_inliers,
                         float angular_threshold, float distance_threshold, float max_curvature, int depth_mode,
                         float* xyz_out, float* normals_out, int32_t* labels_out, rgbd360_plane* planes_out, int max_planes,
                         int* n_planes_out);

/* One sensor's organised cloud as Frame360::buildSphereCloud_rgbd360 hands it to the plane extraction (Frame360.h:479-481):
 * CloudRGBD::getPointCloud (OpenNI2_Grabber/FrameRGBD/CloudRGBD.h:107-166: pinhole, focal 525 * cols / 640, centre (cols/2 - 0.5,
 * rows/2 - 0.5)) followed by DownsampleRGBD::downsamplePointCloud (DownsampleRGBD.h:209-300: per step x step block and per
 * coordinate the element n/2 of the sorted valid values; step 1 = the plain cloud, DOWNSAMPLE_160 = 2; at most 4).  depth: uint16 mm,
 * host, depth_step bytes per row; a pixel is valid when it has a depth and min_depth < z < max_depth (metres; DownsampleRGBD's
 * defaults 0.3 / 10 -- the reference's CloudRGBD.h:133-150 compares metres with millimetre thresholds, which is not reproduced).
 * xyz_out: (rows/step) * (cols/step) x 3 float32, NaN = invalid. */
int rgbd360_sensor_cloud(rgbd360_ctx* ctx, const uint16_t* depth, size_t depth_step, int rows, int cols, int step, float min_depth,
                         float max_depth, float* xyz_out);
/* The same from either kind of sensor image: depth_type 0 = uint16 millimetres (as above), 1 = float32 METRES -- the image
 * Frame360::undistort leaves (CloudRGBD_Ext.h:61-75, 116-118: m_depthEigUndistort), whose points getPointCloudUndist keeps when
 * z > 0 && min_depth <= z <= max_depth (inclusive, in metres: :118) before DownsampleRGBD takes the medians. */
int rgbd360_sensor_cloud_ex(rgbd360_ctx* ctx, const void* depth, size_t depth_step, int depth_type, int rows, int cols, int step,
                            float min_depth, float max_depth, float* xyz_out);

/* Frame360::getPlanesSensor for one organised sensor cloud (host, rows*cols x 3 float32, NaN = invalid), with the smoothing
 * that precedes it: pcl::FastBilateralFilter when sigma_s > 0 (Frame360.h:493-499: 10, 0.05), the normal map (Frame360.h:949-957:
 * 0.02, 8; depth_mode 0 = PCL's z), the planar regions (Frame360.h:958-996: 80 inliers, 0.0398 rad, 0.02 m) -- the cloud stays on
 * the device between the three stages -- and, when Rt (column-major 4x4, Calib360::Rt_, sensor -> rig) is not NULL, the
 * plane.transform(Rt) of Frame360.h:1046.  Planes in PCL's order; n towards the (new) origin, n . x + d = 0. */
int rgbd360_cloud_planes(rgbd360_ctx* ctx, const float* xyz, int rows, int cols, float sigma_s, float sigma_r,
                         float max_depth_change_factor, float normal_smoothing_size, int min_inliers, float angular_threshold,
                         float distance_threshold, float max_curvature, int depth_mode, const float Rt[16],
                         rgbd360_plane* planes_out, int max_planes, int* n_planes_out);

/* rgbd360_sensor_cloud + rgbd360_cloud_planes (depth_mode 0) as one call: the depth image is uploaded once and the cloud never
 * leaves the device -- one sensor of Frame360::buildSphereCloud_rgbd360 + getPlanesSensor (Frame360.h:479-499, 949-996, 1046). */
int rgbd360_sensor_planes(rgbd360_ctx* ctx, const uint16_t* depth, size_t depth_step, int rows, int cols, int step, float min_depth,
                          float max_depth, float sigma_s, float sigma_r, float max_depth_change_factor, float normal_smoothing_size,
                          int min_inliers, float angular_threshold, float distance_threshold, float max_curvature, const float Rt[16],
                          rgbd360_plane* planes_out, int max_planes, int* n_planes_out);
/* rgbd360_sensor_planes from either kind of sensor image (depth_type as in rgbd360_sensor_cloud_ex). */
int rgbd360_sensor_planes_ex(rgbd360_ctx* ctx, const void* depth, size_t depth_step, int depth_type, int rows, int cols, int step,
                             float min_depth, float max_depth, float sigma_s, float sigma_r, float max_depth_change_factor,
                             float normal_smoothing_size, int min_inliers, float angular_threshold, float distance_threshold,
                             float max_curvature, const float Rt[16], rgbd360_plane* planes_out, int max_planes, int* n_planes_out);

/* ---- the sensors' intrinsic depth model: Frame360::undistort (Frame360.h:293-311, undistortDepthSensor :1084-1097) ------------
 * Calib360::loadIntrinsicCalibration (Calib360.h:104-119) loads one clams::DiscreteDepthDistortionModel per sensor from
 * Calibration/Intrinsics/distortion_model<N> and calls downsampleParams(2); Frame360::undistort applies it to each sensor's depth image
 * in metres before the clouds are built.  The model (CLAMS, Teichman et al.; vendored by the reference under
 * OpenNI2_Grabber/third_party/CLAMS) cuts the image into bins of pixels, each with a multiplier per depth slice; z becomes z * m with m
 * interpolated between the two slices around z when both saw >= 50 training examples (rgbd360_amd/csrc/depth_model.h restates file
 * layout and arithmetic).  Host only, no context needed.
 *   load:      0 ok, 1 cannot open, 2 not a model file / truncated / bins not divisible by `downsample`, -1 bad arguments.
 *   info:      dims = {width, height, bin_width, bin_height, num_bins_x, num_bins_y} after the down-sampling.
 *   undistort: rows x cols float32 metres, in place (0 = no measurement stays 0); the image must have the model's size (-1 otherwise). */
typedef struct rgbd360_depth_model rgbd360_depth_model;
int  rgbd360_depth_model_load(const char* path, int downsample, rgbd360_depth_model** out);
void rgbd360_depth_model_free(rgbd360_depth_model* model);
int  rgbd360_depth_model_info(const rgbd360_depth_model* model, int dims[6], double* bin_depth);
int  rgbd360_depth_model_undistort(const rgbd360_depth_model* model, float* depth_m, size_t depth_step, int rows, int cols);

/* ---- pinhole single-sensor alignment ------------------------------------------------------ */

/* RegisterPhotoICP::setCameraMatrix (RPI.h:254-257): fx, fy, ox, oy of the full-resolution sensor image; the pyramid
 * levels scale them by 2^-level (RPI.h:571-575). */
int rgbd360_set_camera(rgbd360_ctx* ctx, float fx, float fy, float ox, float oy);
/* RegisterPhotoICP::alignFrames(pose_guess, method, occlusion) (RPI.h:4254-4512): coarse-to-fine alignment of two
 * pinhole RGB-D images (the frames given to rgbd360_set_target / _source of a context created with mask_seams = 0) with
 * the reference's Levenberg-Marquardt schedule (lambda 0.01, x10 / /10, 10 iterations, tolerances 1e-4 hard-coded at
 * RPI.h:4303-4308), errorPhotoICP (RPI.h:560-748) and calcHessGrad (RPI.h:754-1104), CPose3D::exp (full exponential).
 * The per-pixel passes run on the device, the damping loop on the host.  Reference quirks kept (occlusion 0): the error
 * divides both residual sums by the number of depth-valid pixels, so PHOTO_CONSISTENCY alone gives NaN and the guess comes
 * back (status RGBD360_NO_VALID_PIXELS); the error pass applies no saliency test while the H,g pass does.
 * occlusion 1 / 2 select errorPhotoICP_Occ1 / calcHessGrad_Occ1 (RPI.h:1107-1544) and errorPhotoICP_Occ2 /
 * calcHessGrad_Occ2 (RPI.h:1547-2030) in the SEQUENTIAL semantics of their source (index order; the OpenMP loops race on the
 * z-buffer), as written: Occ2's error gate compares the target depth with the point's INVERSE depth, and both H,g variants
 * sum a pixel's depth row only where its photometric residual is non-zero (DEPTH_CONSISTENCY alone: H = 0, ILL-POSED).
 * No application of the reference calls them (MethodsRegisterRGBD360.cpp:348 passes 0). */
int rgbd360_align_pinhole(rgbd360_ctx* ctx, const float guess[16], int method, int occlusion, float pose_out[16],
                          rgbd360_result* res);
/* RegisterPhotoICP::useSaliency(bool) (RPI.h:266-269) with thresSaliency (RPI.h:217: 0.01): the pinhole error pass
 * (occlusion 0) sums over vSalientPixels only -- the interior pixels whose TARGET gray gradient exceeds the threshold in x
 * or y (calcGradientXY_saliency RPI.h:401-425), used as SOURCE pixel indices (RPI.h:590-690, as written); H, g keep every
 * pixel (their salient branch is commented out, RPI.h:813-870).  No other path of this library reads the list. */
int rgbd360_use_saliency(rgbd360_ctx* ctx, int on, float thres_saliency);
/* One fused pinhole pass at `pose` (stage-level, for parity tests): error sums / counts of errorPhotoICP and H, g of
 * calcHessGrad; n_rows = Jacobian rows that entered the normal equations. */
int rgbd360_eval_pinhole(rgbd360_ctx* ctx, int level, const float pose[16], int method, double err2_split[2],
                         long long n_split[2], float H[36], float g[6], double H64[36], double g64[6], long long* n_rows);
/* The same with the occlusion mode (0: identical to rgbd360_eval_pinhole): error sums / counts of errorPhotoICP_Occ1/2 and
 * H, g of calcHessGrad_Occ1/2 at `pose`; n_rows = numVisiblePixels as the reference counts it (a target pixel's first
 * arrival counts twice, RPI.h:1421-1430). */
int rgbd360_eval_pinhole_occ(rgbd360_ctx* ctx, int level, const float pose[16], int method, int occlusion, double err2_split[2],
                             long long n_split[2], float H[36], float g[6], double H64[36], double g64[6], long long* n_rows);
int rgbd360_warp_indices_pinhole(rgbd360_ctx* ctx, int level, const float pose[16], int32_t* host_out_rc);

/* The same chain with the depth image already in HBM and the maps left there: *xyz_dev, *normals_dev (rows*cols*3 floats) and
 * *labels_dev (rows*cols int32; root pixel index of the region, -1 for invalid points) point into buffers owned by the context,
 * valid until its next Frame360 call; only the plane list comes back to the host (0.33 ms of kernels at 2048x1024 against
 * 8-10 ms when 56 MB of maps travel to pageable host memory). */
int rgbd360_frame_planes_dev(rgbd360_ctx* ctx, const void* depth_dev, size_t depth_step, int depth_type, int rows, int cols,
                             int convention, float max_depth_change_factor, float normal_smoothing_size, int min_inliers,
                             float angular_threshold, float distance_threshold, float max_curvature, int depth_mode,
                             rgbd360_plane* planes_out, int max_planes, int* n_planes_out, const float** xyz_dev,
                             const float** normals_dev, const int32_t** labels_dev);

/* ---- Frame360 input side (the two steps before the path) ------------------------------------------------------- */

/* Frame360::loadFrame (Frame360.h:231-266): reads one `sphere_images_%d.bin` (Boost binary archive of 8 x {RGB 8UC3,
 * depth 16UC1 mm} cv::Mat records).  Host-only.  rgb_out: 8*rows*cols*3 bytes, depth_out: 8*rows*cols uint16; pass
 * NULL buffers to query rows / cols first.  Returns 0, -2 cannot open, -3 truncated, -4 unexpected record. */
int rgbd360_load_frame_bin(const char* path, uint8_t* rgb_out, uint16_t* depth_out, int* rows, int* cols);

/* Frame360::stitchSphericalImage (Frame360.h:386-405, stitchImage :1099-1148): the 8 sensor images (contiguous
 * [8][sensor_rows][sensor_cols][3] uint8 and [8][sensor_rows][sensor_cols] uint16 mm, host) -> panorama of
 * W = sensor_rows*8 columns and H = int(W*0.5*60/180) rows (sphere_rgb_out H*W*3, sphere_depth_out H*W range in mm).
 * Rt_inv: 8 column-major 4x4 (Calib360::Rt_inv), K = {fx, fy, cx, cy} (Calib360.h:74-77). */
int rgbd360_stitch_sphere(rgbd360_ctx* ctx, const uint8_t* rgb8, const uint16_t* depth8, int sensor_rows, int sensor_cols,
                          const float Rt_inv[128], const float K[4], uint8_t* sphere_rgb_out, uint16_t* sphere_depth_out,
                          int* out_rows, int* out_cols);

/* ---- PbMap plane registration: the initial-guess provider in front of the path -------------- */

/* Thresholds of mrpt::pbmap::SubgraphMatcher (config_files/configLocaliser_spherical.ini / ..._sphericalOdometry.ini,
 * loaded at RegisterRGBD360.h:97-100) that the geometric and colour constraints below use. */
typedef struct {
    /* [unary] */
    float dist_d;                 /* odometry modes: |d_ref - d_trg| below this (m) */
    float angle_deg;              /* odometry modes: angle between the two normals below this (deg) */
    float elongation_threshold;   /* ratio of elongations below this */
    float area_threshold;         /* ratio of areas below this */
    /* [binary] */
    float dist_threshold;         /* ratio of the two centre distances below this */
    float angle_threshold_deg;    /* difference of the two inter-normal angles below this (deg) */
    float height_threshold;       /* difference of the perpendicular offsets of one centre over the other plane (m) */
    float cos_normal_threshold;   /* after the fit: n_ref . (R n_trg) of every matched pair above this */
    /* [global] */
    int   min_planes_recognition; /* fewer matches than this = "Insuficient matching" (RegisterRGBD360.h:312-316) */
    float max_curvature_plane;    /* planes above it never enter the subgraphs (RegisterRGBD360.h:121-150; Miscellaneous.h:54) */
    float min_area_plane;         /* Frame360.h:1034 drops smaller planes before they reach the PbMap (Miscellaneous.h:57) */
    float max_elongation_plane;   /* Frame360.h:1041 drops narrower planes (Miscellaneous.h:60) */
    /* planar modes: the rig moves on the floor; normals keep their component along this axis (0 = x, up in the sphere
     * frame of RPI.h:4567-4582) within planar_normal_tol, and horizontal planes keep d within dist_d */
    int   up_axis;
    float planar_normal_tol;
    /* pose fit */
    float max_conditioning;       /* largest / smallest eigenvalue of sum w n n^T above this = ill-conditioned translation */
    float sigma_dist, sigma_normal; /* scale of the information matrix: st. dev. of a plane offset (m) / normal (rad) */
    int   max_nodes;              /* budget of the interpretation-tree search (nodes); 0 = unlimited */
    /* [unary], colour (applied to a pair of planes that both carry colour, color_count > 0) */
    int   use_color;              /* 0: no colour test at all */
    float color_threshold;        /* |difference| of every channel of color_nrgb below this (ini: 0.07) */
    float intensity_threshold;    /* |difference| of intensity below this (ini: 150 on the 0..765 scale); <= 0: not tested */
    float hue_threshold;          /* Bhattacharyya distance sqrt(1 - sum sqrt(h1 h2)) of the two hist_h below this (ini: 0.45); <= 0: not
                                   * tested -- the default: the reference's own use of it is commented out (Frame360.h:673) */
} rgbd360_pbmap_params;

/* odometry = 0: configLocaliser_spherical.ini, 1: configLocaliser_sphericalOdometry.ini */
void rgbd360_pbmap_default_params(rgbd360_pbmap_params* p, int odometry);

/* RegisterRGBD360::RegisterPbMap (RegisterRGBD360.h:276-338): subgraph selection (setReference / setTarget :110-195:
 * planes under max_curvature_plane; if max_match_planes > 0 and there are more, the max_match_planes largest areas),
 * interpretation-tree matching of the two plane sets under unary + binary geometric constraints
 * (mrpt::pbmap::SubgraphMatcher::compareSubgraphs -- third-party; restated from the published method, Fernandez-Moral
 * et al., "Fast place recognition with plane-based maps", ICRA 2013), then the closed-form pose of the matched planes
 * with its information matrix (mrpt::pbmap::ConsistencyTest::estimatePoseWithCovariance -- third-party, restated:
 * rotation = SVD of sum w n_ref n_trg^T, translation = least squares on the plane offsets).  Host code, no device work.
 *   regist_mode: 0 DEFAULT_6DoF, 1 PLANAR_3DoF, 2 ODOMETRY_6DoF, 3 PLANAR_ODOMETRY_3DoF (RegisterRGBD360.h:258-264).
 *   pose_out: column-major 4x4, pose of the target frame seen from the reference, p_ref = R p_trg + t.
 *   info_out: column-major 6x6 information matrix in [t; w] order (left perturbation, like RPI.h:4697).
 *   match_out: n_ref entries, index of the matched target plane or -1.  area_matched_out: matched area in the
 *   reference frame (calcAreaMatched).  Any output pointer may be NULL.
 * Returns 0 good alignment, 1 insufficient matching (< min_planes_recognition; pose_out = identity),
 * 2 ill-conditioned (rotation or translation not observable from the matched normals) or inconsistent fit, -1 bad arguments. */
int rgbd360_register_planes(const rgbd360_plane* ref, int n_ref, const rgbd360_plane* trg, int n_trg, int max_match_planes,
                            int regist_mode, const rgbd360_pbmap_params* params, float pose_out[16], float info_out[36],
                            int32_t* match_out, int* n_matched_out, float* area_matched_out);

/* Frame360::mergePlanes (Frame360.h:655-733): the pieces several sensors (or several regions) hold of one surface become one
 * plane.  Same surface = the reference's explicit test: n_j . n_k > cos_normal (0.99), |d_j - d_k| < dist_d (0.45 m), and outlines
 * closer than proximity (0.3 m) at a pair of points whose difference lies within normal_offset (0.06 m) of plane j -- on the HULL
 * POLYGONS of the two records, as the reference does (Frame360.h:680-691 vertex against vertex, :694-711 edge against edge: the 3-D
 * segment-to-segment distance; round 6: no containment test -- the reference has none, a panel inside a wall's hull but farther than
 * `proximity` from its outline stays a plane of its own).  Records without a polygon (hull_n = 0: caller-made) are tested on the rectangle with their in-plane moments instead (corners,
 * edge midpoints, centre).  The merged plane is the exact pooled fit of the two pieces (covariances rebuilt from the records, combined
 * by inlier count -- mrpt's mergePlane2 pools the inliers and refits); its polygon is the convex hull of the two polygons' vertices
 * projected onto the pooled plane (mergePlane2 re-hulls the two contours), its area and centre that polygon's.
 * Planes above max_curvature are never merged (Frame360.h:659-661).  Regions smaller than min_area (0.12 m2) or narrower than
 * max_elongation (6) are dropped first: Frame360.h:1034,1041 never stores them, so the reference's merge never sees them (and the
 * record of a thin strip pins its normal too loosely to be pooled); malformed records are dropped too.  Host only.
 * out may not alias planes; returns -1 when max_out is too small (*n_out = needed). */
int rgbd360_merge_planes(const rgbd360_plane* planes, int n, float max_curvature, float min_area, float max_elongation, float cos_normal,
                         float dist_d, float proximity, float normal_offset, rgbd360_plane* out, int max_out, int* n_out);

/* Frame360::groupPlanes (Frame360.h:741-833), the step of Frame360::getPlanes (:615-639) between the eight getPlanesSensor calls and
 * mergePlanes: `planes` holds the sensors' plane lists one after the other (n_per_sensor[s] records of sensor s, in the rig frame:
 * rgbd360_sensor_planes / rgbd360_cloud_planes with that sensor's Rt), the output is the frame's list.  A plane of sensor s is pooled
 * (mergePlane2, as in rgbd360_merge_planes) into a plane that came from -- or absorbed a piece of -- sensor s - 1 when both are larger
 * than min_area (0.5 m2) and flatter than max_curvature (0.0013) as the source tests them (:764: area OR curvature for the new piece,
 * :770: area AND curvature for the absorbing one), n_j . n_k > cos_normal (0.99), |d_j - d_k| < dist_d (0.45 m) and their hull polygons
 * come within max_dist_hull (0.5 m) at a pair of points / edges whose difference lies within max_dist_parallel_hull (0.09 m) of the
 * absorbing plane (:788-815); otherwise it is appended.  The last sensor's candidates include the first sensor's planes (the ring of
 * sensors closes, :827-828).  Nothing is filtered here.  Host only.  out may not alias planes; -1 when max_out is too small
 * (*n_out = needed) or on bad arguments. */
int rgbd360_group_planes(const rgbd360_plane* planes, const int* n_per_sensor, int n_sensors, float max_curvature, float min_area,
                         float cos_normal, float dist_d, float max_dist_hull, float max_dist_parallel_hull, rgbd360_plane* out, int max_out,
                         int* n_out);

/* The tail of Frame360::getPlanesSensor (Frame360.h:1034-1068), between one sensor's regions (rgbd360_sensor_planes / _cloud_planes, already
 * in the rig frame) and local_planes_[sensor]: regions smaller than min_area (0.12 m2, :1034) or narrower than max_elongation (6, :1041)
 * are dropped; each remaining region flatter than max_curvature (0.0013) is pooled (mergePlane2, as in rgbd360_merge_planes) into the
 * first plane already kept, also flatter, that mrpt::pbmap::Plane::isSamePlane(plane, cos_normal 0.99, dist_normal 0.05 m,
 * proximity 0.2 m) accepts (:1056-1068) -- normals, the centres' distance along the kept plane's normal, and the outlines nearer than
 * proximity (centres, vertices, edges; MRPT, third-party: restated, unpinned) -- else appended, in input order.  Host only.
 * out may not alias planes; -1 when max_out is too small (*n_out = needed) or on bad arguments. */
int rgbd360_pool_sensor_planes(const rgbd360_plane* planes, int n, float max_curvature, float min_area, float max_elongation, float cos_normal,
                               float dist_normal, float proximity, rgbd360_plane* out, int max_out, int* n_out);

#ifdef __cplusplus
}
#endif
#endif /* RGBD360_HIP_H */
