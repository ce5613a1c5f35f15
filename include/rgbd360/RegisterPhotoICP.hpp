// RegisterPhotoICP.hpp -- C++ adapter over the C ABI (include/rgbd360_hip.h) with the public surface of the
// reference's RegisterPhotoICP for the spherical path (include/RegisterPhotoICP.h "RPI.h" of EduFdez/rgbd360):
// same method names, argument meaning and error behaviour, so call sites such as OdometryRGBD360.cpp:189-193,
// OdometryKeyFrame360.cpp:244-253, LoopClosure360.h:306-321 compile against it unchanged apart from the include.
//
// The adapter is header-only and depends on nothing but the C ABI.  When Eigen / OpenCV headers are present the class has
// the reference's own signatures -- setTargetFrame(cv::Mat&, cv::Mat&), alignFrames360(Eigen::Matrix4f, costFuncType, int),
// Eigen::Matrix4f getOptimalPose(), Eigen::Matrix<float,6,6> getHessian(), Eigen::Matrix<float,6,1> getGradient()
// (RPI.h:273-288, 480-516, 4519) -- so the reference's call sites compile against it character for character
// (tests/test_cpp_adapter.py compiles OdometryRGBD360.cpp:189-193, LoopClosure360.h:306-323 and KFsphere_SLAM.cpp:398-402 in
// place against mock Eigen / OpenCV headers; include/rgbd360/compat/RegisterPhotoICP.h puts the class into the global
// namespace under the reference's header name).  Without them (the build container has neither) the same names return the
// POD Mat4f / Mat6f types below; the POD forms are always available as getOptimalPosePod() / getHessianPod() / getGradientPod().
#pragma once

#include <array>
#include <cmath>
#include <utility>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../rgbd360_hip.h"

#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define RGBD360_HAVE_EIGEN 1
#endif
#if __has_include(<opencv2/core/core.hpp>)
#include <opencv2/core/core.hpp>
#define RGBD360_HAVE_OPENCV 1
#endif
#endif

namespace rgbd360 {

// Column-major 4x4 / 6x6 PODs: the memory layout of Eigen::Matrix4f and Eigen::Matrix<float,6,6>.
struct Mat4f {
    float m[16];
    static Mat4f Identity() {
        Mat4f I{};
        I.m[0] = I.m[5] = I.m[10] = I.m[15] = 1.f;
        return I;
    }
    float& operator()(int r, int c) { return m[c * 4 + r]; }
    float operator()(int r, int c) const { return m[c * 4 + r]; }
};
struct Mat6f {
    float m[36];
    float operator()(int r, int c) const { return m[c * 6 + r]; }
};

// What the path needs from a cv::Mat: data pointer, size, row stride, element type.
struct ImageView {
    const void* data = nullptr;
    int rows = 0, cols = 0;
    size_t step = 0;
    enum Type { U8C3, U16C1, F32C1 } type = U8C3;
};

class RegisterPhotoICP {
   public:
    enum costFuncType { PHOTO_CONSISTENCY, DEPTH_CONSISTENCY, PHOTO_DEPTH };   // RPI.h:194

    // public fields of the reference (RPI.h:177-189); num_iterations (RPI.h:177) is filled by every alignment
    std::vector<int> num_iterations;
    float SSO = 0.f;
    float avResidual = 0.f;
    double avPhotoResidual = 0.0, avDepthResidual = 0.0;
    int nPyrLevels = 4;

    RegisterPhotoICP() { rgbd360_default_params(&p_); }            // RPI.h:201-221
    ~RegisterPhotoICP() { reset(); }
    RegisterPhotoICP(const RegisterPhotoICP&) = delete;
    RegisterPhotoICP& operator=(const RegisterPhotoICP&) = delete;

    // RPI.h:224-269.  Like the reference these are meant to be called before the frames are set.
    void setNumPyr(int Npyr) { nPyrLevels = p_.n_pyr = Npyr; reset(); }
    void setMinDepth(float minD) { p_.min_depth = minD; reset(); }
    void setMaxDepth(float maxD) { p_.max_depth = maxD; reset(); }
    void setGrayVariance(float stdDev) { p_.sigma_photo = stdDev; reset(); }     // sets the std-dev (RPI.h:242-245)
    void setDepthVariance(float stdDev) { p_.sigma_depth = stdDev; reset(); }
    // RPI.h:266-269: selects calcGradientXY_saliency's pixel list (RPI.h:401-425), which only the pinhole error pass reads
    // (RPI.h:590-690); the spherical passes' salient branch is commented out (RPI.h:2568-2642): no effect on alignFrames360.
    void useSaliency(bool on) {
        use_saliency_ = on;
        if (ctx_ && rgbd360_use_saliency(ctx_, on ? 1 : 0, 0.01f) != 0) throw std::runtime_error(std::string("rgbd360_use_saliency: ") + rgbd360_last_error(ctx_));
    }
    void setVisualization(bool viz) {
        if (viz) throw std::runtime_error("rgbd360: visualisation is not part of the MI355X path");
    }
    void setDevice(int device) { p_.device = device; reset(); }
    // RPI.h:254-257 (pinhole single-sensor use).  The seam mask of the spherical panorama (RPI.h:4538-4549) does not apply
    // to a sensor image: alignFrames() needs setMaskSeams(false), which the reference gets implicitly by never running
    // alignFrames360 on such an object.
    void setCameraMatrix(float fx, float fy, float ox, float oy) {
        cam_[0] = fx; cam_[1] = fy; cam_[2] = ox; cam_[3] = oy;
        have_cam_ = true;
        if (ctx_ && rgbd360_set_camera(ctx_, fx, fy, ox, oy) != 0) throw std::runtime_error(std::string("rgbd360_set_camera: ") + rgbd360_last_error(ctx_));
    }
    void setMaskSeams(bool on) { p_.mask_seams = on ? 1 : 0; reset(); }
    /*! Not in the reference: the spherical warp in the reference's OWN arithmetic (asinf / atan2f / roundf as glibc computes them, Eigen's
     *  product order) instead of the device definition -- target indices bit-equal to a CPU build of the reference, the pass ~1.8 x
     *  slower (rgbd360_set_index_arithmetic). */
    void setReferenceArithmetic(bool on) {
        index_libm_ = on ? 1 : 0;
        if (ctx_ && rgbd360_set_index_arithmetic(ctx_, index_libm_) != 0) throw std::runtime_error(std::string("rgbd360_set_index_arithmetic: ") + rgbd360_last_error(ctx_));
    }

    // RPI.h:498-516 / 480-494
    void setTargetFrame(const ImageView& rgb, const ImageView& depth) { set(true, rgb, depth); }
    void setSourceFrame(const ImageView& rgb, const ImageView& depth) { set(false, rgb, depth); }

    // The `setTargetFrame(frame2 ...)` of the next odometry step (OdometryRGBD360.cpp:189) without a second upload: the source
    // frame's pyramids become the target's (rgbd360_promote_source_to_target); a new source must be set afterwards.
    void promoteSourceToTarget() {
        if (rgbd360_promote_source_to_target(ctx()) != 0)
            throw std::runtime_error(std::string("rgbd360_promote_source_to_target: ") + rgbd360_last_error(ctx_));
    }

    // RPI.h:4519-4784.  void like the reference; `status()` tells ill-posed (1) / no valid pixels (2).
#ifdef RGBD360_HAVE_EIGEN      // (the defaults live on the Eigen overloads below, as in the reference)
    void alignFrames360(const Mat4f& pose_guess, costFuncType method = PHOTO_CONSISTENCY, int occlusion = 0) {
#else
    void alignFrames360(const Mat4f& pose_guess = Mat4f::Identity(), costFuncType method = PHOTO_CONSISTENCY,
                        int occlusion = 0) {
#endif
        rgbd360_result r;
        const int rc = rgbd360_align360(ctx(), pose_guess.m, (int)method, occlusion, relPose_.m, &r);
        if (rc < 0) throw std::runtime_error(std::string("rgbd360_align360: ") + rgbd360_last_error(ctx_));
        status_ = rc;
        std::memcpy(hessian_.m, r.hessian, sizeof(hessian_.m));
        std::memcpy(gradient_.data(), r.gradient, sizeof(float) * 6);
        SSO = r.sso;
        avResidual = rc == RGBD360_ILL_POSED ? 0.f : (float)r.err_final;      // RPI.h:4688
        avPhotoResidual = r.rms_photo;
        avDepthResidual = r.rms_depth;
        num_iterations.assign(r.iters, r.iters + p_.n_pyr);
    }

    // RPI.h:4254-4512: pinhole single-sensor alignment (Levenberg-Marquardt); occlusion 1 / 2 = errorPhotoICP_Occ1/2 +
    // calcHessGrad_Occ1/2 (RPI.h:1107-2030) in their sequential semantics.
#ifdef RGBD360_HAVE_EIGEN
    void alignFrames(const Mat4f& pose_guess, costFuncType method = PHOTO_CONSISTENCY, int occlusion = 0) {
#else
    void alignFrames(const Mat4f& pose_guess = Mat4f::Identity(), costFuncType method = PHOTO_CONSISTENCY, int occlusion = 0) {
#endif
        rgbd360_result r;
        const int rc = rgbd360_align_pinhole(ctx(), pose_guess.m, (int)method, occlusion, relPose_.m, &r);
        if (rc < 0) throw std::runtime_error(std::string("rgbd360_align_pinhole: ") + rgbd360_last_error(ctx_));
        status_ = rc;
        std::memcpy(hessian_.m, r.hessian, sizeof(hessian_.m));
        std::memcpy(gradient_.data(), r.gradient, sizeof(float) * 6);
        if (occlusion == 2) SSO = r.sso;            // only calcHessGrad_Occ2 sets it on this path (RPI.h:2016)
        avResidual = (float)r.err_final;
        avPhotoResidual = r.rms_photo;
        avDepthResidual = r.rms_depth;
        num_iterations.assign(r.iters, r.iters + p_.n_pyr);
    }

    // The frame loop of OdometryRGBD360.cpp:141-297 as one call (rgbd360_align360_batch): pair j aligns frame j+1 (source)
    // to frame j (target); every frame is uploaded once, n_inflight pairs are in flight on the GPU (lock-step slots of the sequence engine).  All
    // frames must share one size and depth type, and stay untouched until the call returns.  Returns the relative poses;
    // statuses (0 / ILL_POSED / NO_VALID_PIXELS per pair) and full records through the optional outputs.
    std::vector<Mat4f> alignSequence(const std::vector<ImageView>& rgb, const std::vector<ImageView>& depth,
                                     costFuncType method = PHOTO_DEPTH, int occlusion = 0, int n_inflight = 32,
                                     const Mat4f& pose_guess = Mat4f::Identity(), std::vector<rgbd360_result>* results = nullptr) {
        if (rgb.size() != depth.size()) throw std::invalid_argument("rgbd360: rgb / depth sequence length mismatch");
        const size_t n_frames = rgb.size();
        std::vector<Mat4f> poses(n_frames > 0 ? n_frames - 1 : 0, Mat4f::Identity());
        if (results) results->assign(poses.size(), rgbd360_result{});
        if (poses.empty()) return poses;
        std::vector<const uint8_t*> prgb(n_frames);
        std::vector<const void*> pdepth(n_frames);
        for (size_t k = 0; k < n_frames; ++k) {
            if (rgb[k].type != ImageView::U8C3 || depth[k].type == ImageView::U8C3 || depth[k].type != depth[0].type ||
                rgb[k].rows != rgb[0].rows || rgb[k].cols != rgb[0].cols || depth[k].rows != rgb[0].rows || depth[k].cols != rgb[0].cols ||
                rgb[k].step != rgb[0].step || depth[k].step != depth[0].step)
                throw std::invalid_argument("rgbd360: all frames of a sequence must share one size, step and depth type");
            prgb[k] = (const uint8_t*)rgb[k].data;
            pdepth[k] = depth[k].data;
        }
        static_assert(sizeof(Mat4f) == 16 * sizeof(float), "Mat4f must be 16 packed floats");
        const int rc = rgbd360_align360_batch(ctx(), (int)n_frames, prgb.data(), rgb[0].step, pdepth.data(), depth[0].step,
                                              depth[0].type == ImageView::U16C1 ? 0 : 1, rgb[0].rows, rgb[0].cols, pose_guess.m,
                                              (int)method, occlusion, n_inflight, poses[0].m, results ? results->data() : nullptr);
        if (rc < 0) throw std::runtime_error(std::string("rgbd360_align360_batch: ") + rgbd360_last_error(ctx_));
        return poses;
    }

    // results as PODs (always), and under the reference's names: Eigen types when Eigen is there (below), else the PODs
    Mat4f getOptimalPosePod() const { return relPose_; }
    Mat6f getHessianPod() const { return hessian_; }
    std::array<float, 6> getGradientPod() const { return gradient_; }
#ifndef RGBD360_HAVE_EIGEN
    Mat4f getOptimalPose() const { return relPose_; }          // RPI.h:273
    Mat6f getHessian() const { return hessian_; }              // RPI.h:279
    std::array<float, 6> getGradient() const { return gradient_; }     // RPI.h:285
#endif
    const std::vector<int>& numIterations() const { return num_iterations; }
    int status() const { return status_; }

    // RPI.h:4789-4797 (call site OdometryRGBD360.cpp:207): differential entropy of the pose estimate, 0.5 (DOF (1 + ln 2 pi) + ln det H^-1)
    // with DOF = 6 and H the Hessian of the last alignment.  ln det H^-1 = -ln det H from a partial-pivot LU in double: the reference
    // forms det(H.inverse()) in float, which underflows to 0 (entropy -inf) once det H passes 3e38 -- a 2048 x 1024 Hessian does; the
    // value here stays finite and equals the reference's wherever that one is finite.  NaN when H is singular or not positive.
    float calcEntropy() const {
        double a[6][6], logdet = 0;
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) a[r][c] = hessian_.m[c * 6 + r];
        for (int k = 0; k < 6; ++k) {
            int p = k;
            for (int r = k + 1; r < 6; ++r)
                if (std::fabs(a[r][k]) > std::fabs(a[p][k])) p = r;
            if (p != k)
                for (int c = 0; c < 6; ++c) std::swap(a[k][c], a[p][c]);
            logdet += std::log(std::fabs(a[k][k]));          // (a symmetric positive definite H has a positive determinant: the sign is dropped)
            if (a[k][k] == 0) return std::nanf("");
            for (int r = k + 1; r < 6; ++r) {
                const double f = a[r][k] / a[k][k];
                for (int c = k; c < 6; ++c) a[r][c] -= f * a[k][c];
            }
        }
        const double PI_ = 3.14159265359;                    // the reference's macro (Miscellaneous.h:44)
        return (float)(0.5 * (6.0 * (1.0 + std::log(2 * PI_)) - logdet));
    }

#ifdef RGBD360_HAVE_OPENCV
    void setTargetFrame(cv::Mat& imgRGB, cv::Mat& imgDepth) { set(true, view(imgRGB), view(imgDepth)); }
    void setSourceFrame(cv::Mat& imgRGB, cv::Mat& imgDepth) { set(false, view(imgRGB), view(imgDepth)); }
    static ImageView view(const cv::Mat& m) {
        ImageView v;
        v.data = m.data; v.rows = m.rows; v.cols = m.cols; v.step = m.step;
        v.type = m.type() == CV_8UC3 ? ImageView::U8C3 : (m.type() == CV_16UC1 ? ImageView::U16C1 : ImageView::F32C1);
        return v;
    }
#endif
#ifdef RGBD360_HAVE_EIGEN
    void setCameraMatrix(Eigen::Matrix3f& camMat) { setCameraMatrix(camMat(0, 0), camMat(1, 1), camMat(0, 2), camMat(1, 2)); }     // RPI.h:254-257
    // RPI.h:4254 / 4519: by-value Eigen guess with the reference's defaults
    void alignFrames(const Eigen::Matrix4f pose_guess = Eigen::Matrix4f::Identity(), costFuncType method = PHOTO_CONSISTENCY, int occlusion = 0) {
        Mat4f g;
        std::memcpy(g.m, pose_guess.data(), sizeof(g.m));
        alignFrames(g, method, occlusion);
    }
    void alignFrames360(const Eigen::Matrix4f pose_guess = Eigen::Matrix4f::Identity(), costFuncType method = PHOTO_CONSISTENCY, int occlusion = 0) {
        Mat4f g;
        std::memcpy(g.m, pose_guess.data(), sizeof(g.m));
        alignFrames360(g, method, occlusion);
    }
    // RPI.h:273-288: the reference's return types (column-major storage on both sides, hence plain copies)
    Eigen::Matrix4f getOptimalPose() const {
        Eigen::Matrix4f T;
        std::memcpy(T.data(), relPose_.m, sizeof(relPose_.m));
        return T;
    }
    Eigen::Matrix<float, 6, 6> getHessian() const {
        Eigen::Matrix<float, 6, 6> H;
        std::memcpy(H.data(), hessian_.m, sizeof(hessian_.m));
        return H;
    }
    Eigen::Matrix<float, 6, 1> getGradient() const {
        Eigen::Matrix<float, 6, 1> g;
        std::memcpy(g.data(), gradient_.data(), sizeof(float) * 6);
        return g;
    }
    Eigen::Matrix4f getOptimalPoseEigen() const { return getOptimalPose(); }                 // (round-2 names, kept)
    Eigen::Matrix<float, 6, 6> getHessianEigen() const { return getHessian(); }
    // RPI.h:171: LUT_xyz_sphere of pyramid level `level` as the reference keeps it (a download: the alignment itself never needs it on the host)
    std::vector<Eigen::Vector3f> LUT_xyz_sphere;
    void downloadLUT(int level) {
        int rows = 0, cols = 0;
        if (rgbd360_level_dims(ctx(), level, &rows, &cols) != 0) throw std::runtime_error(std::string("rgbd360_level_dims: ") + rgbd360_last_error(ctx_));
        std::vector<float> xyz((size_t)rows * cols * 3);
        if (rgbd360_get_lut(ctx_, level, xyz.data()) != 0) throw std::runtime_error(std::string("rgbd360_get_lut: ") + rgbd360_last_error(ctx_));
        LUT_xyz_sphere.resize((size_t)rows * cols);
        for (size_t i = 0; i < LUT_xyz_sphere.size(); ++i) std::memcpy(LUT_xyz_sphere[i].data(), &xyz[3 * i], 3 * sizeof(float));
    }
#endif
#ifdef RGBD360_HAVE_OPENCV
    // RPI.h:198-199: the public pyramids (CV_32FC1 per level).  They live in HBM; downloadPyramids() fills the vectors for
    // callers that read them (RegisterRGBD360.h:385-388 does; the alignment itself never needs them on the host).
    std::vector<cv::Mat> graySrcPyr, grayTrgPyr, depthSrcPyr, depthTrgPyr, grayTrgGradXPyr, grayTrgGradYPyr, depthTrgGradXPyr, depthTrgGradYPyr;
    void downloadPyramids() {
        std::vector<cv::Mat>* dst[8] = {&graySrcPyr, &grayTrgPyr, &depthSrcPyr, &depthTrgPyr, &grayTrgGradXPyr, &grayTrgGradYPyr, &depthTrgGradXPyr, &depthTrgGradYPyr};
        for (int which = 0; which < 8; ++which) {
            dst[which]->resize((size_t)p_.n_pyr);
            for (int level = 0; level < p_.n_pyr; ++level) {
                int rows = 0, cols = 0;
                if (rgbd360_level_dims(ctx(), level, &rows, &cols) != 0) throw std::runtime_error(std::string("rgbd360_level_dims: ") + rgbd360_last_error(ctx_));
                cv::Mat& m = (*dst[which])[(size_t)level];
                m.create(rows, cols, CV_32FC1);
                if (rgbd360_get_plane(ctx_, which, level, reinterpret_cast<float*>(m.data)) != 0)
                    throw std::runtime_error(std::string("rgbd360_get_plane: ") + rgbd360_last_error(ctx_));
            }
        }
    }
#endif

    // The library context behind this object (created on first use) for the calls that have no RegisterPhotoICP counterpart:
    // the Frame360 stages (rgbd360_frame_planes ...) share its stream and device buffers.
    rgbd360_ctx* context() { return ctx(); }

   private:
    rgbd360_params p_;
    rgbd360_ctx* ctx_ = nullptr;
    int index_libm_ = 0;
    Mat4f relPose_ = Mat4f::Identity();
    Mat6f hessian_{};
    std::array<float, 6> gradient_{};
    int status_ = 0;
    float cam_[4] = {0.f, 0.f, 0.f, 0.f};
    bool have_cam_ = false;
    bool use_saliency_ = false;

    void reset() {
        if (ctx_) rgbd360_destroy(ctx_);
        ctx_ = nullptr;
    }
    rgbd360_ctx* ctx() {
        if (!ctx_) {
            const int rc = rgbd360_create(&p_, &ctx_);
            if (rc != 0) throw std::runtime_error("rgbd360_create failed (" + std::to_string(rc) + "): no usable HIP device; there is no CPU fallback");
            if (have_cam_) rgbd360_set_camera(ctx_, cam_[0], cam_[1], cam_[2], cam_[3]);
            if (use_saliency_) rgbd360_use_saliency(ctx_, 1, 0.01f);
            if (index_libm_) rgbd360_set_index_arithmetic(ctx_, index_libm_);
        }
        return ctx_;
    }
    void set(bool target, const ImageView& rgb, const ImageView& depth) {
        if (rgb.type != ImageView::U8C3) throw std::invalid_argument("rgbd360: imgRGB must be CV_8UC3");
        if (depth.type == ImageView::U8C3) throw std::invalid_argument("rgbd360: imgDepth must be CV_16UC1 (mm) or CV_32FC1 (m)");
        if (rgb.rows != depth.rows || rgb.cols != depth.cols) throw std::invalid_argument("rgbd360: rgb / depth size mismatch");
        const int dt = depth.type == ImageView::U16C1 ? 0 : 1;
        const int rc = (target ? rgbd360_set_target : rgbd360_set_source)(ctx(), (const uint8_t*)rgb.data, rgb.step, depth.data,
                                                                         depth.step, dt, rgb.rows, rgb.cols);
        if (rc != 0) throw std::runtime_error(std::string("rgbd360_set_frame: ") + rgbd360_last_error(ctx_));
    }
};

// The literal north-star shape `bool Register(Frame360&, Frame360&, Eigen::Matrix4f&)`: FrameLike is anything with
// public `sphereRGB` / `sphereDepth` image members (Frame360.h:104-111); `pose` carries the initial guess in and the
// solved relative pose out.
template <class FrameLike, class ToView>
bool Register(FrameLike& trg, FrameLike& src, Mat4f& pose, ToView to_view,
              RegisterPhotoICP::costFuncType method = RegisterPhotoICP::PHOTO_DEPTH, RegisterPhotoICP* reg = nullptr) {
    RegisterPhotoICP local;
    RegisterPhotoICP& r = reg ? *reg : local;
    r.setTargetFrame(to_view(trg.sphereRGB), to_view(trg.sphereDepth));
    r.setSourceFrame(to_view(src.sphereRGB), to_view(src.sphereDepth));
    r.alignFrames360(pose, method);
    pose = r.getOptimalPosePod();
    return r.status() == 0;
}

#if defined(RGBD360_HAVE_EIGEN) && defined(RGBD360_HAVE_OPENCV)
// ... and with the reference's own types: frames whose sphereRGB / sphereDepth are cv::Mat (Frame360.h:104-111), an Eigen pose.
template <class FrameLike>
bool Register(FrameLike& trg, FrameLike& src, Eigen::Matrix4f& pose,
              RegisterPhotoICP::costFuncType method = RegisterPhotoICP::PHOTO_DEPTH, RegisterPhotoICP* reg = nullptr) {
    RegisterPhotoICP local;
    RegisterPhotoICP& r = reg ? *reg : local;
    r.setTargetFrame(trg.sphereRGB, trg.sphereDepth);
    r.setSourceFrame(src.sphereRGB, src.sphereDepth);
    r.alignFrames360(pose, method);
    pose = r.getOptimalPose();
    return r.status() == 0;
}
#endif

}  // namespace rgbd360
