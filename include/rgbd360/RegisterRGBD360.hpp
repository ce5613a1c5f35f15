// RegisterRGBD360.hpp -- C++ adapter over the C ABI (include/rgbd360_hip.h) with the public surface of the reference's
// RegisterRGBD360 (include/RegisterRGBD360.h:47-338 of EduFdez/rgbd360) for the PbMap registration that provides the
// initial guess of the dense alignment: setReference / setTarget / RegisterPbMap / getPose / getCovMat / getInfoMat /
// calcEntropy / getMatchedPlanes / getAreaMatched, same names, argument meaning and return values, so call sites such as
// SphereGraphSLAM.cpp:180, KFsphere_SLAM.cpp:182,314-317 keep their shape.  A "frame" is anything that exposes its planar
// regions as a contiguous `rgbd360_plane` array (what rgbd360_frame_planes[_dev] returns) -- the only member of Frame360
// this class reads is `planes.vPlanes`.  Header-only; depends on nothing but the C ABI.
#pragma once

#include <algorithm>
#include <cmath>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "RegisterPhotoICP.hpp"

namespace rgbd360 {

struct PlaneList {                       // view of a frame's planes; the caller keeps ownership
    const rgbd360_plane* planes = nullptr;
    int n = 0;
};

// Frame360::segmentPlanes / getPlanes (Frame360.h:615-720, 949-1075) for one range panorama: sphere cloud -> normal map ->
// planar regions on the device (rgbd360_frame_planes), the plane list comes back to the host.  Parameters default to the
// reference's (PCL set-up of Frame360.h:949-977: depth-change factor 0.02, smoothing 8, min_inliers 80, 0.0398 rad, 0.02 m;
// region curvature filter = PCL's default 0.001; max_curvature_plane of Miscellaneous.h:54 only enters at mergePlanes / setReference); convention / depth_mode as in rgbd360_hip.h (2 / 1 = full sphere, range).
struct SegmentParams {
    int convention = 2, depth_mode = 1, min_inliers = 80, max_planes = 256;
    float max_depth_change_factor = 0.02f, normal_smoothing_size = 8.f, angular_threshold = 0.0398f, distance_threshold = 0.02f,
          max_curvature = 0.001f;      // PCL's default region filter: the reference never calls mps.setMaximumCurvature (Frame360.h:958-977)
    bool refine = true;                // the reference calls segmentAndRefine (Frame360.h:977): planes grow into their noisy borders
    float refine_distance = 0.02f;     // PlaneRefinementComparator's default threshold (not depth dependent)
};
// rgb (optional): the frame's colour panorama (8UC3, same size as depth) -- the planes then carry the colour descriptors of
// Frame360.h:1045-1046 (calcPlaneHistH / calcMainColor2), which RegisterPbMap's unary colour constraint reads.
inline std::vector<rgbd360_plane> segmentPlanes(RegisterPhotoICP& reg, const ImageView& depth, const SegmentParams& sp = SegmentParams(),
                                                const ImageView* rgb = nullptr) {
    std::vector<rgbd360_plane> planes;
    int n = 0, cap = sp.max_planes;
    const int dt = depth.type == ImageView::U16C1 ? 0 : 1;
    rgbd360_ctx* ctx = reg.context();
    rgbd360_set_plane_refinement(ctx, sp.refine ? 1 : 0, sp.refine_distance);
    if (rgb && rgb->data && rgb->type == ImageView::U8C3 && rgb->rows == depth.rows && rgb->cols == depth.cols) {
        if (rgbd360_set_plane_color_image(ctx, (const uint8_t*)rgb->data, rgb->step, rgb->rows, rgb->cols, 1, 0) != 0)
            throw std::runtime_error(std::string("rgbd360_set_plane_color_image: ") + rgbd360_last_error(ctx));
    } else {
        rgbd360_set_plane_color_image(ctx, nullptr, 0, 0, 0, 1, 0);
    }
    for (int attempt = 0; attempt < 2; ++attempt) {      // the library keeps the largest `cap` regions and reports how many qualified: grow once
        planes.resize((size_t)cap);
        const int rc = rgbd360_frame_planes(ctx, depth.data, depth.step, dt, depth.rows, depth.cols, sp.convention, sp.max_depth_change_factor,
                                            sp.normal_smoothing_size, sp.min_inliers, sp.angular_threshold, sp.distance_threshold,
                                            sp.max_curvature, sp.depth_mode, nullptr, nullptr, nullptr, planes.data(), cap, &n);
        if (rc != 0) throw std::runtime_error(std::string("rgbd360_frame_planes: ") + rgbd360_last_error(ctx));
        const int avail = rgbd360_planes_available(ctx);
        if (avail <= n) break;
        cap = avail;
    }
    planes.resize((size_t)n);
    return planes;
}

// Frame360::getPlanesSensor for one sensor's organised cloud (rows*cols x 3 float32, NaN = invalid; what CloudRGBD::getPointCloud
// + DownsampleRGBD produce), smoothed by the bilateral filter of Frame360.h:493-499 first, planes moved into the rig frame by
// Rt (Calib360::Rt_[sensor], column-major 4x4; nullptr = leave them in the sensor frame): rgbd360_cloud_planes.
struct SensorSegmentParams {
    float sigma_s = 10.f, sigma_r = 0.05f;                                     // Frame360.h:496-497 (sigma_s <= 0: no filter)
    float max_depth_change_factor = 0.02f, normal_smoothing_size = 8.f;         // Frame360.h:952-953
    int min_inliers = 80;                                                        // Frame360.h:960
    float angular_threshold = 0.0398f, distance_threshold = 0.02f;               // Frame360.h:961-962
    float max_curvature = 0.001f;                                                // PCL default (no setMaximumCurvature call); 0.0013 = max_curvature_plane is the merge / subgraph filter
    int max_planes = 512;
    bool refine = true;                // segmentAndRefine (Frame360.h:977)
    float refine_distance = 0.02f;
    // the tail of getPlanesSensor (Frame360.h:1034-1068; Miscellaneous.h:54,57,60)
    float max_curvature_plane = 0.0013f, min_area_plane = 0.12f, max_elongation_plane = 6.f;
};
inline std::vector<rgbd360_plane> segmentSensorPlanes(RegisterPhotoICP& reg, const float* xyz, int rows, int cols, const float* Rt = nullptr,
                                                      const SensorSegmentParams& sp = SensorSegmentParams()) {
    std::vector<rgbd360_plane> planes;
    int n = 0, cap = sp.max_planes;
    rgbd360_ctx* ctx = reg.context();
    rgbd360_set_plane_refinement(ctx, sp.refine ? 1 : 0, sp.refine_distance);
    for (int attempt = 0; attempt < 2; ++attempt) {      // grow once when more regions qualified than the buffer holds
        planes.resize((size_t)cap);
        const int rc = rgbd360_cloud_planes(ctx, xyz, rows, cols, sp.sigma_s, sp.sigma_r, sp.max_depth_change_factor, sp.normal_smoothing_size,
                                            sp.min_inliers, sp.angular_threshold, sp.distance_threshold, sp.max_curvature, /*depth_mode=*/0, Rt,
                                            planes.data(), cap, &n);
        if (rc != 0) throw std::runtime_error(std::string("rgbd360_cloud_planes: ") + rgbd360_last_error(ctx));
        const int avail = rgbd360_planes_available(ctx);
        if (avail <= n) break;
        cap = avail;
    }
    planes.resize((size_t)n);
    return planes;
}

// Frame360::mergePlanes (Frame360.h:655-733; defaults = its constants): the pieces several sensors hold of one surface become one plane.
inline std::vector<rgbd360_plane> mergePlanes(const std::vector<rgbd360_plane>& planes, float max_curvature = 0.0013f, float cos_normal = 0.99f,
                                              float dist_d = 0.45f, float proximity = 0.3f, float normal_offset = 0.06f, float min_area = 0.12f,
                                              float max_elongation = 6.f) {
    std::vector<rgbd360_plane> out(planes.size() ? planes.size() : 1);
    int n = 0;
    if (rgbd360_merge_planes(planes.data(), (int)planes.size(), max_curvature, min_area, max_elongation, cos_normal, dist_d, proximity, normal_offset, out.data(),
                             (int)out.size(), &n) != 0)
        throw std::runtime_error("rgbd360_merge_planes: bad arguments");
    out.resize((size_t)n);
    return out;
}

// The tail of Frame360::getPlanesSensor (Frame360.h:1034-1068; defaults = its constants): one sensor's regions -> local_planes_[sensor]:
// regions under min_area_plane / over max_elongation_plane dropped, flat regions of one surface (isSamePlane(0.99, 0.05, 0.2)) pooled.
inline std::vector<rgbd360_plane> poolSensorPlanes(const std::vector<rgbd360_plane>& planes, float max_curvature = 0.0013f, float min_area = 0.12f,
                                                   float max_elongation = 6.f, float cos_normal = 0.99f, float dist_normal = 0.05f, float proximity = 0.2f) {
    std::vector<rgbd360_plane> out(planes.size() ? planes.size() : 1);
    int n = 0;
    if (rgbd360_pool_sensor_planes(planes.data(), (int)planes.size(), max_curvature, min_area, max_elongation, cos_normal, dist_normal, proximity, out.data(),
                                   (int)out.size(), &n) != 0)
        throw std::runtime_error("rgbd360_pool_sensor_planes: bad arguments");
    out.resize((size_t)n);
    return out;
}

// Frame360::groupPlanes (Frame360.h:741-833; defaults = its constants): the eight sensors' plane lists (rig frame, sensor order) -> the
// frame's list, pieces of one surface seen by neighbouring sensors pooled.  getPlanes (:615-639) = the sensors' planes, this, mergePlanes.
inline std::vector<rgbd360_plane> groupPlanes(const std::vector<std::vector<rgbd360_plane>>& per_sensor, float max_curvature = 0.0013f, float min_area = 0.5f,
                                              float cos_normal = 0.99f, float dist_d = 0.45f, float max_dist_hull = 0.5f, float max_dist_parallel_hull = 0.09f) {
    std::vector<rgbd360_plane> all;
    std::vector<int> counts;
    for (const std::vector<rgbd360_plane>& v : per_sensor) {
        counts.push_back((int)v.size());
        all.insert(all.end(), v.begin(), v.end());
    }
    std::vector<rgbd360_plane> out(all.size() ? all.size() : 1);
    int n = 0;
    if (rgbd360_group_planes(all.data(), counts.data(), (int)counts.size(), max_curvature, min_area, cos_normal, dist_d, max_dist_hull, max_dist_parallel_hull,
                             out.data(), (int)out.size(), &n) != 0)
        throw std::runtime_error("rgbd360_group_planes: bad arguments");
    out.resize((size_t)n);
    return out;
}

class RegisterRGBD360 {
   public:
    enum registrationType { DEFAULT_6DoF, PLANAR_3DoF, ODOMETRY_6DoF, PLANAR_ODOMETRY_3DoF };      // RegisterRGBD360.h:258-264

    float areaSource = 0.f, areaTarget = 0.f;                                                      // :88-92

    // :97-105 loads config_files/configLocaliser_spherical*.ini; here the two parameter sets are built in
    explicit RegisterRGBD360(bool odometry_config = false) { rgbd360_pbmap_default_params(&params_, odometry_config ? 1 : 0); }
    explicit RegisterRGBD360(const rgbd360_pbmap_params& p) : params_(p) {}

    rgbd360_pbmap_params& params() { return params_; }

    // The planes are COPIED (a few KB): the reference keeps Frame360 pointers whose owners outlive the registration; a view into
    // a caller's temporary vector would dangle as soon as a later setReference / getPose re-runs the registration.
    void setReference(const PlaneList& ref, size_t max_match_planes = 0) {       // :110-157
        ref_own_.assign(ref.planes, ref.planes + (ref.planes ? ref.n : 0));
        ref_ = PlaneList{ref_own_.data(), (int)ref_own_.size()};
        max_ref_ = max_match_planes;
        done_ = false;
    }
    void setTarget(const PlaneList& trg, size_t max_match_planes = 0) {          // :163-195
        trg_own_.assign(trg.planes, trg.planes + (trg.planes ? trg.n : 0));
        trg_ = PlaneList{trg_own_.data(), (int)trg_own_.size()};
        max_trg_ = max_match_planes;
        done_ = false;
    }

    // :276-338.  true = good alignment; false = "Insuficient matching" or an unobservable / inconsistent fit.
    bool RegisterPbMap(const PlaneList* frame1 = nullptr, const PlaneList* frame2 = nullptr, size_t max_match_planes = 0,
                       registrationType registMode = DEFAULT_6DoF) {
        if (frame1) setReference(*frame1, max_match_planes);
        if (frame2) setTarget(*frame2, max_match_planes);
        mode_ = registMode;
        done_ = true;
        std::vector<int32_t> match(ref_.n > 0 ? ref_.n : 1, -1);
        int n_matched = 0;
        float pose[16], info[36];
        const size_t mmp = max_ref_ > max_trg_ ? max_ref_ : max_trg_;
        status_ = rgbd360_register_planes(ref_.planes, ref_.n, trg_.planes, trg_.n, (int)mmp, (int)registMode, &params_, pose, info,
                                          match.data(), &n_matched, &areaMatched_);
        bestMatch_.clear();
        for (int i = 0; i < ref_.n; ++i)
            if (match[i] >= 0) bestMatch_[(unsigned)i] = (unsigned)match[i];
        if (status_ != 0) return false;
        std::memcpy(rigidTransf_.m, pose, sizeof(pose));
        std::memcpy(informationM_.m, info, sizeof(info));
        areaSource = subgraphArea(ref_, mmp);                                                     // :325-333
        areaTarget = subgraphArea(trg_, mmp);
        return true;
    }

    // The reference's signatures take Frame360 pointers (RegisterRGBD360.h:110, 163, 276): anything with a `planes.vPlanes` vector of
    // plane records (rgbd360::Frame360 of Frame360.hpp) goes through the PlaneList forms above.
    template <class FrameT, class = decltype(std::declval<FrameT&>().planes.vPlanes.data())>
    void setReference(FrameT* frame, size_t max_match_planes = 0) {
        setReference(PlaneList{frame->planes.vPlanes.data(), (int)frame->planes.vPlanes.size()}, max_match_planes);
    }
    template <class FrameT, class = decltype(std::declval<FrameT&>().planes.vPlanes.data())>
    void setTarget(FrameT* frame, size_t max_match_planes = 0) {
        setTarget(PlaneList{frame->planes.vPlanes.data(), (int)frame->planes.vPlanes.size()}, max_match_planes);
    }
    template <class FrameT, class = decltype(std::declval<FrameT&>().planes.vPlanes.data())>
    bool RegisterPbMap(FrameT* frame1, FrameT* frame2, size_t max_match_planes = 0, registrationType registMode = DEFAULT_6DoF) {
        const PlaneList a{frame1->planes.vPlanes.data(), (int)frame1->planes.vPlanes.size()}, b{frame2->planes.vPlanes.data(), (int)frame2->planes.vPlanes.size()};
        return RegisterPbMap(&a, &b, max_match_planes, registMode);
    }

    Mat4f getPose() {                                                            // :198-204
        ensure();
        return rigidTransf_;
    }
    Mat6f& getInfoMat() {                                                        // :218-224
        ensure();
        return informationM_;
    }
    Mat6f getCovMat() {                                                          // :207-215
        ensure();
        Mat6f c{};
        double det;
        invert6(informationM_, c, det);
        return c;
    }
    float calcEntropy() {                                                        // :229-238
        ensure();
        Mat6f c{};
        double det_info;
        invert6(informationM_, c, det_info);
        const double pi = 3.14159265358979323846;
        return (float)(0.5 * (6 * (1 + std::log(2 * pi)) + std::log(1.0 / det_info)));
    }
    std::map<unsigned, unsigned> getMatchedPlanes() {                            // :241-247
        ensure();
        return bestMatch_;
    }
    float getAreaMatched() {                                                     // :250-256
        ensure();
        return areaMatched_;
    }
    int status() const { return status_; }      // 0 good, 1 insufficient matching, 2 unobservable / inconsistent

    // :344-520 RegisterDensePhotoICP(frame1, frame2, pose_estim, method): dense registration of the two frames' 8 sensor image pairs
    // (frame->frameRGBD_[s].getRGBImage() / getDepthImage()) in the rig frame, rgbd360_rig_* (csrc/rig_dense.h).  Rt: the sensors'
    // sensor -> rig poses (frame1->calib->Rt_); the intrinsics are the reference's 525 * width / 640, centre (:357-365).  The
    // reference function is broken as written; the library implements it with the three fixes documented in rgbd360_hip.h.
    // true: registered (getPose() = rigidTransf, getInfoMat() = the summed Hessian); false: "The problem is ILL-POSED".
    bool RegisterDensePhotoICP(const std::vector<ImageView>& rgb1, const std::vector<ImageView>& depth1, const std::vector<ImageView>& rgb2,
                               const std::vector<ImageView>& depth2, const std::vector<Mat4f>& Rt, Mat4f pose_estim = Mat4f::Identity(),
                               RegisterPhotoICP::costFuncType method = RegisterPhotoICP::PHOTO_CONSISTENCY, int n_pyr = 4) {
        const size_t S = Rt.size();
        if (S == 0 || rgb1.size() != S || depth1.size() != S || rgb2.size() != S || depth2.size() != S)
            throw std::runtime_error("RegisterDensePhotoICP: one image pair and one extrinsic per sensor");
        const int rows = rgb1[0].rows, cols = rgb1[0].cols;
        const float focal = 525.f * ((float)cols / 640.f);
        rgbd360_params p;
        rgbd360_default_params(&p);
        p.n_pyr = n_pyr;
        std::vector<float> rt(16 * S);
        for (size_t s = 0; s < S; ++s)
            for (int k = 0; k < 16; ++k) rt[16 * s + k] = Rt[s].m[k];
        // every view is validated BEFORE the rig exists, and the handle (a sequence engine with its device buffers, pinned memory and
        // streams) is owned by a unique_ptr from the moment it does: nothing below can leak it
        for (const std::vector<ImageView>* set_of : {&rgb1, &depth1, &rgb2, &depth2})
            for (size_t s = 0; s < S; ++s) {
                const ImageView& v = (*set_of)[s];
                const ImageView& first = (*set_of)[0];
                if (v.rows != rows || v.cols != cols || v.step != first.step || v.type != first.type || !v.data)
                    throw std::runtime_error("RegisterDensePhotoICP: all sensor images must share one size, stride and depth type");
            }
        rgbd360_rig* rig_raw = nullptr;
        if (rgbd360_rig_create(&p, (int)S, rt.data(), focal, focal, cols / 2.f - 0.5f, rows / 2.f - 0.5f, &rig_raw) != 0)
            throw std::runtime_error("rgbd360_rig_create failed: no usable HIP device (there is no CPU fallback)");
        const std::unique_ptr<rgbd360_rig, void (*)(rgbd360_rig*)> rig_owner(rig_raw, rgbd360_rig_destroy);
        rgbd360_rig* rig = rig_raw;
        auto set = [&](bool target, const std::vector<ImageView>& rgb, const std::vector<ImageView>& depth) {
            std::vector<const uint8_t*> rp(S);
            std::vector<const void*> dp(S);
            for (size_t s = 0; s < S; ++s) {
                if (rgb[s].rows != rows || rgb[s].cols != cols || depth[s].rows != rows || depth[s].cols != cols || rgb[s].step != rgb[0].step ||
                    depth[s].step != depth[0].step || depth[s].type != depth[0].type)
                    throw std::runtime_error("RegisterDensePhotoICP: all sensor images must share one size, stride and depth type");
                rp[s] = (const uint8_t*)rgb[s].data;
                dp[s] = depth[s].data;
            }
            const int dt = depth[0].type == ImageView::U16C1 ? 0 : 1;
            return target ? rgbd360_rig_set_target(rig, rp.data(), rgb[0].step, dp.data(), depth[0].step, dt, rows, cols)
                          : rgbd360_rig_set_source(rig, rp.data(), rgb[0].step, dp.data(), depth[0].step, dt, rows, cols);
        };
        int rc = set(true, rgb1, depth1);
        if (rc == 0) rc = set(false, rgb2, depth2);
        rgbd360_result res;
        if (rc == 0) rc = rgbd360_rig_align(rig, pose_estim.m, (int)method, rigidTransf_.m, &res);
        if (rc < 0) throw std::runtime_error(std::string("RegisterDensePhotoICP: ") + rgbd360_rig_last_error(rig));
        for (int k = 0; k < 36; ++k) informationM_.m[k] = res.hessian[k];
        done_ = true;                                   // bRegistrationDone = true   (:506)
        status_ = rc;
        return rc == 0;
    }

   private:
    void ensure() {
        if (!done_) RegisterPbMap(nullptr, nullptr, 0, mode_);
    }
    // area of the planes that entered the matching (:121-150, :325-333): the library's subgraph selection restated
    float subgraphArea(const PlaneList& f, size_t max_match) const {
        std::vector<float> areas;
        for (int i = 0; i < f.n; ++i) {
            const rgbd360_plane& p = f.planes[i];
            if (p.area < params_.min_area_plane || p.elongation > params_.max_elongation_plane) continue;
            areas.push_back(p.curvature < params_.max_curvature_plane ? p.area : 0.f);
        }
        float thr = -1.f;
        if (max_match > 0 && areas.size() > max_match) {
            std::vector<float> sorted = areas;
            std::sort(sorted.begin(), sorted.end());
            thr = sorted[areas.size() - max_match - 1];
        }
        float a = 0.f;
        for (float v : areas)
            if (v > thr && v > 0.f) a += v;
        return a;
    }
    // Gauss-Jordan with partial pivoting in double; det = determinant of the input
    static void invert6(const Mat6f& in, Mat6f& out, double& det) {
        double a[6][12];
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) {
                a[r][c] = in(r, c);
                a[r][6 + c] = r == c;
            }
        det = 1;
        for (int k = 0; k < 6; ++k) {
            int p = k;
            for (int r = k + 1; r < 6; ++r)
                if (std::fabs(a[r][k]) > std::fabs(a[p][k])) p = r;
            if (p != k) {
                for (int c = 0; c < 12; ++c) std::swap(a[k][c], a[p][c]);
                det = -det;
            }
            det *= a[k][k];
            const double inv = 1.0 / a[k][k];
            for (int c = 0; c < 12; ++c) a[k][c] *= inv;
            for (int r = 0; r < 6; ++r) {
                if (r == k) continue;
                const double f = a[r][k];
                for (int c = 0; c < 12; ++c) a[r][c] -= f * a[k][c];
            }
        }
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) out.m[c * 6 + r] = (float)a[r][6 + c];
    }

    rgbd360_pbmap_params params_{};
    std::vector<rgbd360_plane> ref_own_, trg_own_;       // owned copies; ref_ / trg_ view them
    PlaneList ref_{}, trg_{};
    size_t max_ref_ = 0, max_trg_ = 0;
    registrationType mode_ = DEFAULT_6DoF;
    bool done_ = false;
    int status_ = 1;
    Mat4f rigidTransf_ = Mat4f::Identity();
    Mat6f informationM_{};
    std::map<unsigned, unsigned> bestMatch_;
    float areaMatched_ = 0.f;
};

// The keyframe link of the reference's SLAM applications as one call (KFsphere_SLAM.cpp:129-163 with :182 / :314 in front):
// planes of both frames -> RegisterPbMap -> alignFrames360 seeded with the plane pose (with `pose` as passed in when the plane
// registration fails) -> the reference's validity test `dense.isApprox(pbmap, 1e-1)` when both exist.  `pose` carries the
// fallback guess in and the dense pose out (p_ref = R p_cur + t).  FrameLike: public `sphereRGB` / `sphereDepth` members.
template <class FrameLike, class ToView>
bool RegisterFrames(FrameLike& ref, FrameLike& cur, Mat4f& pose, ToView to_view, RegisterPhotoICP& dense, RegisterRGBD360& registerer,
                    RegisterRGBD360::registrationType registMode = RegisterRGBD360::ODOMETRY_6DoF, size_t max_match_planes = 25,
                    const SegmentParams& seg = SegmentParams(),
                    RegisterPhotoICP::costFuncType method = RegisterPhotoICP::PHOTO_DEPTH) {
    const ImageView rgb_r = to_view(ref.sphereRGB), rgb_c = to_view(cur.sphereRGB);
    const std::vector<rgbd360_plane> pr = segmentPlanes(dense, to_view(ref.sphereDepth), seg, &rgb_r);
    const std::vector<rgbd360_plane> pc = segmentPlanes(dense, to_view(cur.sphereDepth), seg, &rgb_c);
    PlaneList lr{pr.data(), (int)pr.size()}, lc{pc.data(), (int)pc.size()};
    const bool planes_ok = registerer.RegisterPbMap(&lr, &lc, max_match_planes, registMode);
    const Mat4f guess = planes_ok ? registerer.getPose() : pose;
    dense.setTargetFrame(to_view(ref.sphereRGB), to_view(ref.sphereDepth));
    dense.setSourceFrame(to_view(cur.sphereRGB), to_view(cur.sphereDepth));
    dense.alignFrames360(guess, method);
    pose = dense.getOptimalPosePod();
    if (dense.status() != 0) return false;
    if (planes_ok) {                                     // Eigen's isApprox(b, p): ||a - b|| <= p min(||a||, ||b||), Frobenius
        double diff = 0, na = 0, nb = 0;
        for (int k = 0; k < 16; ++k) {
            diff += ((double)pose.m[k] - guess.m[k]) * ((double)pose.m[k] - guess.m[k]);
            na += (double)pose.m[k] * pose.m[k];
            nb += (double)guess.m[k] * guess.m[k];
        }
        if (diff > 1e-2 * (na < nb ? na : nb)) return false;
    }
    return true;
}

}  // namespace rgbd360
