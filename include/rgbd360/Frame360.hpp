// Frame360.hpp -- C++ adapter over the C ABI (include/rgbd360_hip.h) with the public surface of the reference's Frame360 and
// Calib360 (include/Frame360.h:93-1148, include/Calib360.h:44-134 of EduFdez/rgbd360) for the stages this library runs on the
// device: loadFrame, undistort (host), fastStitchImage360 (host), stitchSphericalImage, buildSphereCloud, buildSphereCloud_fromImage, getPlanes (= the eight getPlanesSensor calls, groupPlanes,
// mergePlanes), getPlanesSensor, segmentPlanes (the one-panorama variant of Frame360_stereo.h:835-980), getPlanarArea,
// getAverageIntensity -- same member and method names, so call sites such as RegisterPairRGBD360.cpp:95-110 or
// OdometryRGBD360.cpp:150-176 (`frame.loadFrame(file); frame.stitchSphericalImage(); frame.getPlanes(); ... frame.planes.vPlanes`)
// keep their shape.  Images are owned byte buffers viewed through ImageView (the adapter has no OpenCV dependency; with OpenCV a
// cv::Mat header over the same memory is one line).  NOT mirrored: what depends on third-party file formats or models that are not in
// the reference tree (loadCloud / loadPbMap / save / serialize: PCL and MRPT serialisation).
// Frame360_stereo (include/Frame360_stereo.h) follows at the end.  Header-only; depends on nothing but the C ABI and the two other adapter headers.
#pragma once

#include <array>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "RegisterRGBD360.hpp"

namespace rgbd360 {

// Calib360.h:44-134: the rig's extrinsics Rt_[8] (sensor -> rig), their inverses and the sensors' pinhole matrix.
class Calib360 {
   public:
    enum Resolution { VGA = 1, QVGA = 2, QQVGA = 4 } resolution;      // Calib360.h:61-66
    std::array<Mat4f, 8> Rt_, Rt_inv;
    float cameraMatrix[9];                                            // row-major 3x3, Calib360.h:74-77 (QVGA)

    explicit Calib360(Resolution res = QVGA) : resolution(res) {
        const float K[9] = {262.5f, 0.f, 159.5f, 0.f, 262.5f, 119.5f, 0.f, 0.f, 1.f};
        for (int i = 0; i < 9; ++i) cameraMatrix[i] = K[i];
        for (int s = 0; s < 8; ++s) Rt_[s] = Rt_inv[s] = Mat4f::Identity();
    }
    Mat4f getRt_id(int id) const { return Rt_.at((size_t)id); }
    void setRt_id(int id, const Mat4f& Rt) {
        Rt_.at((size_t)id) = Rt;
        Rt_inv.at((size_t)id) = rigid_inverse(Rt);
    }
    // Calib360.h:122-131: Rt_0N.txt, N = 1..8, plain text 4x4 (row by row).  Returns false when a file is missing or short
    // (the reference's loadFromTextFile throws).
    bool loadExtrinsicCalibration(const std::string& pathToExtrinsicModel) {
        for (int s = 0; s < 8; ++s) {
            char name[16];
            std::snprintf(name, sizeof(name), "/Rt_0%d.txt", s + 1);
            std::FILE* f = std::fopen((pathToExtrinsicModel + name).c_str(), "r");
            if (!f) return false;
            Mat4f M{};
            bool ok = true;
            for (int r = 0; r < 4 && ok; ++r)
                for (int c = 0; c < 4 && ok; ++c) ok = std::fscanf(f, "%f", &M(r, c)) == 1;
            std::fclose(f);
            if (!ok) return false;
            setRt_id(s, M);
        }
        return true;
    }
    // Calib360.h:104-119: the sensors' intrinsic depth models, Calibration/Intrinsics/distortion_model<N> (N = 1..8), prepared for the
    // half-resolution images the rig delivers (downsampleParams(2)).  false when a file is missing or not a model (the source asserts).
    bool loadIntrinsicCalibration(const std::string& pathToIntrinsicModel) {
        for (int s = 0; s < 8; ++s) {
            rgbd360_depth_model* m = nullptr;
            if (rgbd360_depth_model_load((pathToIntrinsicModel + "/distortion_model" + std::to_string(s + 1)).c_str(), 2, &m) != 0) return false;
            intrinsic_model_[(size_t)s] = std::shared_ptr<rgbd360_depth_model>(m, rgbd360_depth_model_free);
        }
        return true;
    }
    std::array<std::shared_ptr<rgbd360_depth_model>, 8> intrinsic_model_;      // Calib360.h:56

    // {fx, fy, cx, cy} as the C ABI takes them
    std::array<float, 4> K() const { return {cameraMatrix[0], cameraMatrix[4], cameraMatrix[2], cameraMatrix[5]}; }

    // The rig's device side: one context (stream + device buffers) per sensor, created on first use and shared by every Frame360 of this
    // calibration -- a frame object stays as light as the reference's (creating eight contexts costs tens of milliseconds; a frame's
    // planes take two).  Like the reference's classes: one thread works with a rig's frames at a time.
    RegisterPhotoICP& context(int sensor_id) {
        if (!contexts_) contexts_ = std::make_shared<std::array<RegisterPhotoICP, 8>>();
        return contexts_->at((size_t)sensor_id);
    }

   private:
    std::shared_ptr<std::array<RegisterPhotoICP, 8>> contexts_;

   public:
    static Mat4f rigid_inverse(const Mat4f& T) {      // [R t; 0 1]^-1 = [R^T  -R^T t; 0 1]
        Mat4f I = Mat4f::Identity();
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) I(r, c) = T(c, r);
        for (int r = 0; r < 3; ++r) I(r, 3) = -(I(r, 0) * T(0, 3) + I(r, 1) * T(1, 3) + I(r, 2) * T(2, 3));
        return I;
    }
};

// mrpt::pbmap::PbMap as far as the reference's call sites read it: the plane vector.
struct PbMap {
    std::vector<rgbd360_plane> vPlanes;
};

class Frame360 {
   public:
    unsigned id = 0, node = 0;                      // Frame360.h:98-101
    std::vector<uint8_t> sphereRGB_data;            // the panorama of stitchSphericalImage, 8UC3 ...
    std::vector<uint16_t> sphereDepth_data;         // ... and its range image, 16UC1 mm
    ImageView sphereRGB, sphereDepth;               // views of the two (Frame360.h:104-107 holds cv::Mat)
    std::array<PbMap, 8> local_planes_;             // :110: the planes of each sensor (rig frame)
    Mat4f pose = Mat4f::Identity();                 // :117
    std::vector<float> sphereCloud;                 // :120: buildSphereCloud_fromImage's organised cloud, rows*cols x 3 (NaN = invalid)
    PbMap planes;                                   // :123
    Calib360* calib;                                // :126
    uint64_t timeStamp = 0;                         // :131
    SensorSegmentParams sensor_params;              // getPlanesSensor's PCL set-up (Frame360.h:949-977, min_inliers for the 160x120 clouds below)
    SegmentParams sphere_params;                    // segmentPlanes' (one panorama)

    explicit Frame360(Calib360* calib360) : calib(calib360) { sensor_params.min_inliers = 40; }      // a quarter of the points of the 320x240 setting
    Frame360(const Frame360&) = delete;
    Frame360& operator=(const Frame360&) = delete;

    int sensorRows() const { return srows_; }
    int sensorCols() const { return scols_; }
    ImageView sensorRGB(int sensor_id) const {
        ImageView v;
        v.data = rgb8_.data() + (size_t)sensor_id * srows_ * scols_ * 3; v.rows = srows_; v.cols = scols_; v.step = (size_t)scols_ * 3; v.type = ImageView::U8C3;
        return v;
    }
    ImageView sensorDepth(int sensor_id) const {
        ImageView v;
        v.data = depth8_.data() + (size_t)sensor_id * srows_ * scols_; v.rows = srows_; v.cols = scols_; v.step = (size_t)scols_ * 2; v.type = ImageView::U16C1;
        return v;
    }

    void setTimeStamp(uint64_t timestamp) { timeStamp = timestamp; }                                  // :181
    float getPlanarArea() const {                                                                     // :157-164
        float planarArea = 0;
        for (const rgbd360_plane& p : planes.vPlanes) planarArea += p.area;
        return planarArea;
    }
    // :269 (the body is commented out in the source; its intent -- the mean intensity over every sample-th pixel of the 8 sensors, rounded)
    int getAverageIntensity(int sample = 1) const {
        if (sample < 1 || rgb8_.empty()) return 0;
        unsigned long long sum = 0, n = 0;
        for (size_t i = 0; i + 2 < rgb8_.size(); i += (size_t)3 * sample, ++n)
            sum += (4899u * rgb8_[i] + 9617u * rgb8_[i + 1] + 1868u * rgb8_[i + 2] + 8192u) >> 14;      // cv::cvtColor's fixed-point grey
        return n ? (int)((double)sum / (double)n + 0.5) : 0;
    }

    // :231-266: one binary frame file (8 x {RGB, depth}); throws on an unreadable file (the reference prints and returns)
    void loadFrame(const std::string& binaryFile) {
        int rows = 0, cols = 0;
        if (rgbd360_load_frame_bin(binaryFile.c_str(), nullptr, nullptr, &rows, &cols) != 0) throw std::runtime_error("Frame360::loadFrame: cannot read " + binaryFile);
        rgb8_.resize((size_t)8 * rows * cols * 3);
        depth8_.resize((size_t)8 * rows * cols);
        if (rgbd360_load_frame_bin(binaryFile.c_str(), rgb8_.data(), depth8_.data(), &rows, &cols) != 0)
            throw std::runtime_error("Frame360::loadFrame: cannot read " + binaryFile);
        srows_ = rows; scols_ = cols;
        depth_undist_.clear();
    }
    // the same from memory: [8][rows][cols][3] uint8 and [8][rows][cols] uint16 mm
    void setSensorImages(const uint8_t* rgb8, const uint16_t* depth8, int rows, int cols) {
        rgb8_.assign(rgb8, rgb8 + (size_t)8 * rows * cols * 3);
        depth8_.assign(depth8, depth8 + (size_t)8 * rows * cols);
        srows_ = rows; scols_ = cols;
        depth_undist_.clear();
    }

    // :293-311 (undistortDepthSensor :1084-1097): every sensor's depth image in metres (CloudRGBD_Ext.h:64-69: the millimetre image times
    // 0.001) corrected by its intrinsic model; the clouds and planes of this frame are then built from the corrected float images
    // (getPointCloudUndist, CloudRGBD_Ext.h:78-131).  The panorama (stitchSphericalImage) keeps using the sensor's own images, as in the source.
    void undistort() {
        need_images("undistort");
        const size_t n = (size_t)srows_ * scols_;
        depth_undist_.resize(8 * n);
        for (int s = 0; s < 8; ++s) {
            if (!calib->intrinsic_model_[(size_t)s]) throw std::runtime_error("Frame360::undistort: no intrinsic model (Calib360::loadIntrinsicCalibration first)");
            float* z = depth_undist_.data() + (size_t)s * n;
            const uint16_t* d = depth8_.data() + (size_t)s * n;
            for (size_t i = 0; i < n; ++i) z[i] = (float)(0.001 * (double)d[i]);
            if (rgbd360_depth_model_undistort(calib->intrinsic_model_[(size_t)s].get(), z, (size_t)scols_ * 4, srows_, scols_) != 0) {
                depth_undist_.clear();
                throw std::runtime_error("Frame360::undistort: the sensor images are not of the model's size");
            }
        }
    }
    bool undistorted() const { return !depth_undist_.empty(); }
    ImageView sensorDepthUndistorted(int sensor_id) const {      // float32 metres; empty before undistort()
        ImageView v;
        if (depth_undist_.empty()) return v;
        v.data = depth_undist_.data() + (size_t)sensor_id * srows_ * scols_; v.rows = srows_; v.cols = scols_; v.step = (size_t)scols_ * 4; v.type = ImageView::F32C1;
        return v;
    }

    // :347-383: the eight colour images side by side, each transposed and flipped (sensor 7 - k in the k-th strip) -- no calibration, no
    // projection, colour only (sphereDepth is left empty: the source's depth half is commented out).  Host code.
    void fastStitchImage360() {
        need_images("fastStitchImage360");
        const int W = srows_ * 8, H = scols_;
        sphereRGB_data.assign((size_t)H * W * 3, 0);
        for (int k = 0; k < 8; ++k) {
            const uint8_t* src = rgb8_.data() + (size_t)(7 - k) * srows_ * scols_ * 3;
            for (int r = 0; r < H; ++r)
                for (int c = 0; c < srows_; ++c)      // transpose, then flip about the horizontal axis: out(r, c) = src(c, cols - 1 - r)
                    for (int ch = 0; ch < 3; ++ch)
                        sphereRGB_data[((size_t)r * W + (size_t)k * srows_ + c) * 3 + ch] = src[((size_t)c * scols_ + (scols_ - 1 - r)) * 3 + ch];
        }
        sphereRGB.data = sphereRGB_data.data(); sphereRGB.rows = H; sphereRGB.cols = W; sphereRGB.step = (size_t)W * 3; sphereRGB.type = ImageView::U8C3;
        sphereDepth_data.clear();
        sphereDepth = ImageView();
    }

    // :386-405 (stitchImage :1099-1148): the panorama of the eight sensor images through Rt_inv and K
    void stitchSphericalImage() {
        need_images("stitchSphericalImage");
        const int W = srows_ * 8, H = (int)(W * 0.5 * 60.0 / 180);
        sphereRGB_data.assign((size_t)H * W * 3, 0);
        sphereDepth_data.assign((size_t)H * W, 0);
        float Rinv[128];
        for (int s = 0; s < 8; ++s)
            for (int k = 0; k < 16; ++k) Rinv[s * 16 + k] = calib->Rt_inv[(size_t)s].m[k];
        const std::array<float, 4> K = calib->K();
        int orow = 0, ocol = 0;
        if (rgbd360_stitch_sphere(reg(0).context(), rgb8_.data(), depth8_.data(), srows_, scols_, Rinv, K.data(), sphereRGB_data.data(),
                                  sphereDepth_data.data(), &orow, &ocol) != 0)
            throw std::runtime_error(std::string("rgbd360_stitch_sphere: ") + rgbd360_last_error(reg(0).context()));
        sphereRGB.data = sphereRGB_data.data(); sphereRGB.rows = orow; sphereRGB.cols = ocol; sphereRGB.step = (size_t)ocol * 3; sphereRGB.type = ImageView::U8C3;
        sphereDepth.data = sphereDepth_data.data(); sphereDepth.rows = orow; sphereDepth.cols = ocol; sphereDepth.step = (size_t)ocol * 2; sphereDepth.type = ImageView::U16C1;
    }

    // :467-520: the eight sensor clouds (pinhole cloud down-sampled by 2 :471-474, bilateral filter :485-491) moved into the rig frame by Rt_
    // (:493) -> cloud_[sensor] (rows/2 * cols/2 x 3 floats, NaN = invalid) and their concatenation sphereCloud, sensor after sensor.
    // (after undistort() the clouds are those of the corrected float images, as getPointCloudUndist's are in the source.)
    std::array<std::vector<float>, 8> cloud_;
    void buildSphereCloud() {
        need_images("buildSphereCloud");
        const int orow = srows_ / 2, ocol = scols_ / 2;
        const size_t np = (size_t)orow * ocol;
        for (int s = 0; s < 8; ++s) {
            rgbd360_ctx* ctx = reg(s).context();
            std::vector<float>& c = cloud_[(size_t)s];
            c.resize(np * 3);
            const ImageView dv = sensor_depth_in_use(s);
            int rc = rgbd360_sensor_cloud_ex(ctx, dv.data, dv.step, dv.type == ImageView::F32C1 ? 1 : 0, srows_, scols_, 2, 0.3f, 10.f, c.data());
            if (rc == 0 && sensor_params.sigma_s > 0.f) rc = rgbd360_bilateral_filter(ctx, c.data(), orow, ocol, sensor_params.sigma_s, sensor_params.sigma_r, c.data());
            if (rc != 0) throw std::runtime_error("Frame360::buildSphereCloud (sensor " + std::to_string(s) + "): " + rgbd360_last_error(ctx));
            const Mat4f& T = calib->Rt_[(size_t)s];
            for (size_t i = 0; i < np; ++i) {                   // pcl::transformPointCloud; NaN stays NaN
                const float x = c[3 * i], y = c[3 * i + 1], z = c[3 * i + 2];
                c[3 * i] = T(0, 0) * x + T(0, 1) * y + T(0, 2) * z + T(0, 3);
                c[3 * i + 1] = T(1, 0) * x + T(1, 1) * y + T(1, 2) * z + T(1, 3);
                c[3 * i + 2] = T(2, 0) * x + T(2, 1) * y + T(2, 2) * z + T(2, 3);
            }
        }
        sphereCloud.clear();
        for (int s = 0; s < 8; ++s) sphereCloud.insert(sphereCloud.end(), cloud_[(size_t)s].begin(), cloud_[(size_t)s].end());
    }
    const std::vector<float>& getCloud_id(int id) const { return cloud_.at((size_t)id); }      // :167-171

    // :555-612: the organised cloud of the range panorama (convention 0: the rig's 60-degree band; sphereCloud = rows*cols x 3)
    void buildSphereCloud_fromImage() {
        if (!sphereDepth.data) throw std::runtime_error("Frame360::buildSphereCloud_fromImage: no panorama (call stitchSphericalImage first)");
        sphereCloud.resize((size_t)sphereDepth.rows * sphereDepth.cols * 3);
        if (rgbd360_sphere_cloud(reg(0).context(), sphereDepth.data, sphereDepth.step, 0, sphereDepth.rows, sphereDepth.cols, /*convention=*/0, sphereCloud.data()) != 0)
            throw std::runtime_error(std::string("rgbd360_sphere_cloud: ") + rgbd360_last_error(reg(0).context()));
    }

    // :942-1075: the planes of ONE sensor's cloud (pinhole cloud down-sampled by 2, bilateral filter :493-499, normal map, regions with
    // refinement, hull, extent and the colour descriptors of :1045-1046 from the sensor's own image), moved into the rig frame by
    // Rt_[sensor] (:1048), then the source's tail (:1034-1068, round 6): regions under min_area_plane / over max_elongation_plane are
    // never stored, flat regions of one surface are pooled (isSamePlane(0.99, 0.05, 0.2) + mergePlane2) -> local_planes_[sensor_id].
    // (getLocalPlanes / getLocalPlanesInFrame, :641-655 / :839-940, estimate normals with PCL's COVARIANCE_MATRIX method, which this
    // library does not implement: not mirrored.)
    void getPlanesSensor(int sensor_id) {
        need_images("getPlanesSensor");
        const SensorSegmentParams& sp = sensor_params;
        rgbd360_ctx* ctx = reg(sensor_id).context();
        std::vector<rgbd360_plane>& out = local_planes_.at((size_t)sensor_id).vPlanes;
        int n = 0, cap = sp.max_planes;
        int rc = rgbd360_set_plane_refinement(ctx, sp.refine ? 1 : 0, sp.refine_distance);
        // calcPlaneHistH / calcMainColor2 (:1045-1046) read the plane's points' colours: cloud pixel (r, c) of the cloud down-sampled by 2
        // carries the colour of image pixel (2 r + 1, 2 c + 1) (DownsampleRGBD.h:240, 285-287)
        if (rc == 0) {
            const ImageView cv = sensorRGB(sensor_id);
            rc = rgbd360_set_plane_color_image(ctx, (const uint8_t*)cv.data, cv.step, cv.rows, cv.cols, /*step=*/2, /*on_device=*/0);
        }
        for (int attempt = 0; attempt < 2 && rc == 0; ++attempt) {      // grow once when more regions qualified than the buffer holds
            out.resize((size_t)cap);
            const ImageView dv = sensor_depth_in_use(sensor_id);
            rc = rgbd360_sensor_planes_ex(ctx, dv.data, dv.step, dv.type == ImageView::F32C1 ? 1 : 0, srows_, scols_, /*step=*/2, 0.3f, 10.f, sp.sigma_s, sp.sigma_r,
                                          sp.max_depth_change_factor, sp.normal_smoothing_size, sp.min_inliers, sp.angular_threshold, sp.distance_threshold,
                                          sp.max_curvature, calib->Rt_[(size_t)sensor_id].m, out.data(), cap, &n);
            const int avail = rgbd360_planes_available(ctx);
            if (rc != 0 || avail <= n) break;
            cap = avail;
        }
        if (rc != 0) {
            out.clear();
            throw std::runtime_error("rgbd360_sensor_planes (sensor " + std::to_string(sensor_id) + "): " + rgbd360_last_error(ctx));
        }
        out.resize((size_t)n);
        out = rgbd360::poolSensorPlanes(out, sp.max_curvature_plane, sp.min_area_plane, sp.max_elongation_plane);      // :1034-1068
    }
    // :741-833: the sensors' lists -> `planes`, pieces of one surface seen by neighbouring sensors pooled
    void groupPlanes() {
        std::vector<std::vector<rgbd360_plane>> per_sensor;
        for (const PbMap& m : local_planes_) per_sensor.push_back(m.vPlanes);
        planes.vPlanes = rgbd360::groupPlanes(per_sensor);
    }
    // :657-733
    void mergePlanes() { planes.vPlanes = rgbd360::mergePlanes(planes.vPlanes); }
    // :615-639
    void getPlanes() {
        need_images("getPlanes");
        (void)reg(0);      // the calibration object's context array exists before the threads look at it (each then creates its own element)
        // one host thread per sensor like the source's `#pragma omp parallel num_threads(8)` (:619-623): every sensor has its own context
        std::array<std::string, 8> err;
        std::vector<std::thread> workers;
        for (int s = 0; s < 8; ++s)
            workers.emplace_back([&, s]() {
                try {
                    getPlanesSensor(s);
                } catch (const std::exception& e) {
                    err[(size_t)s] = e.what();
                }
            });
        for (std::thread& w : workers) w.join();
        for (int s = 0; s < 8; ++s)
            if (!err[(size_t)s].empty()) throw std::runtime_error(err[(size_t)s]);
        groupPlanes();      // planes detected from adjacent sensors -> `planes`
        mergePlanes();      // merge big planes
    }
    // Frame360_stereo.h:835-980 (segmentPlanes): planes of the PANORAMA itself (sphere cloud -> normal map -> regions on the device), with
    // the panorama's colours for the descriptors of Frame360.h:1045-1046
    void segmentPlanes() {
        if (!sphereDepth.data) throw std::runtime_error("Frame360::segmentPlanes: no panorama (call stitchSphericalImage first)");
        SegmentParams sp = sphere_params;
        sp.convention = 0;
        planes.vPlanes = rgbd360::segmentPlanes(reg(0), sphereDepth, sp, sphereRGB.data ? &sphereRGB : nullptr);
    }

    // the context of sensor s (the calibration object's: shared by the rig's frames; sensor 0's also serves the panorama stages)
    RegisterPhotoICP& reg(int s) { return calib->context(s); }

   private:
    ImageView sensor_depth_in_use(int s) const { return depth_undist_.empty() ? sensorDepth(s) : sensorDepthUndistorted(s); }
    void need_images(const char* who) const {
        if (rgb8_.empty() || srows_ <= 0) throw std::runtime_error(std::string("Frame360::") + who + ": no sensor images (loadFrame / setSensorImages first)");
    }
    std::vector<uint8_t> rgb8_;
    std::vector<uint16_t> depth8_;
    std::vector<float> depth_undist_;      // [8][rows][cols] metres, after undistort()
    int srows_ = 0, scols_ = 0;
};

// Frame360_stereo.h:90-980: the spherical frame of the stereo omnidirectional camera -- one float32 range panorama in metres (the sensor's
// band of the sphere: rows start 166 steps of 2 pi / cols above the equator, :470-490) with an optional colour panorama.  loadDepth reads
// the sensor's raw file (:268-311); loadRGB takes decoded pixels (the reference decodes a PNG through OpenCV, which this header does not
// depend on); buildSphereCloud :454-512; getPlanesStereo :847-980 with its PCL set-up (0.05 depth-change factor, 40 inliers, 0.05 rad, 0.05 m).
class Frame360_stereo {
   public:
    unsigned id = 0, node = 0;
    std::vector<float> sphereDepth_data;            // rows x cols float32 metres, row-major
    std::vector<uint8_t> sphereRGB_data;            // rows x cols x 3 (optional)
    ImageView sphereRGB, sphereDepth;
    std::vector<float> sphereCloud;                 // rows*cols x 3 (NaN = invalid)
    bool bSphereCloudBuilt = false;                 // :143
    PbMap planes;
    Mat4f pose = Mat4f::Identity();
    SegmentParams params;

    Frame360_stereo() {
        params.convention = 1; params.depth_mode = 1;
        params.max_depth_change_factor = 0.05f; params.min_inliers = 40; params.angular_threshold = 0.05f; params.distance_threshold = 0.05f;      // :857-866
    }
    Frame360_stereo(const Frame360_stereo&) = delete;
    Frame360_stereo& operator=(const Frame360_stereo&) = delete;

    // :268-311: uint16 height, uint16 width, then width x height float32 (the image stored column by column)
    void loadDepth(const std::string& binaryDepthFile) {
        std::FILE* f = std::fopen(binaryDepthFile.c_str(), "rb");
        if (!f) throw std::runtime_error("Frame360_stereo::loadDepth: " + binaryDepthFile + " does NOT EXIST");
        uint16_t hw[2] = {0, 0};
        bool ok = std::fread(hw, 2, 2, f) == 2 && hw[0] > 0 && hw[1] > 0;
        const int height = hw[0], width = hw[1];
        std::vector<float> aux;
        if (ok) {
            aux.resize((size_t)height * width);
            ok = std::fread(aux.data(), 4, aux.size(), f) == aux.size();
        }
        std::fclose(f);
        if (!ok) throw std::runtime_error("Frame360_stereo::loadDepth: " + binaryDepthFile + " is short");
        sphereDepth_data.resize((size_t)height * width);
        for (int c = 0; c < width; ++c)              // cv::transpose of the width x height block
            for (int r = 0; r < height; ++r) sphereDepth_data[(size_t)r * width + c] = aux[(size_t)c * height + r];
        sphereDepth.data = sphereDepth_data.data(); sphereDepth.rows = height; sphereDepth.cols = width; sphereDepth.step = (size_t)width * 4; sphereDepth.type = ImageView::F32C1;
        bSphereCloudBuilt = false;
    }
    // :318-340 with the decoding left to the caller: rows x cols x 3 bytes of the size of the range image
    void loadRGB(const uint8_t* rgb, int rows, int cols) {
        sphereRGB_data.assign(rgb, rgb + (size_t)rows * cols * 3);
        sphereRGB.data = sphereRGB_data.data(); sphereRGB.rows = rows; sphereRGB.cols = cols; sphereRGB.step = (size_t)cols * 3; sphereRGB.type = ImageView::U8C3;
    }
    // :454-512
    void buildSphereCloud() {
        if (!sphereDepth.data) throw std::runtime_error("Frame360_stereo::buildSphereCloud: no range image (loadDepth first)");
        sphereCloud.resize((size_t)sphereDepth.rows * sphereDepth.cols * 3);
        if (rgbd360_sphere_cloud(reg_.context(), sphereDepth.data, sphereDepth.step, 1, sphereDepth.rows, sphereDepth.cols, /*convention=*/1, sphereCloud.data()) != 0)
            throw std::runtime_error(std::string("rgbd360_sphere_cloud: ") + rgbd360_last_error(reg_.context()));
        bSphereCloudBuilt = true;
    }
    // :847-980 (cloud, normal map, regions with refinement, descriptors: one chain on the device, from the range image)
    void getPlanesStereo() {
        if (!sphereDepth.data) throw std::runtime_error("Frame360_stereo::getPlanesStereo: no range image (loadDepth first)");
        const bool coloured = sphereRGB.data && sphereRGB.rows == sphereDepth.rows && sphereRGB.cols == sphereDepth.cols;
        planes.vPlanes = rgbd360::segmentPlanes(reg_, sphereDepth, params, coloured ? &sphereRGB : nullptr);
    }
    RegisterPhotoICP& reg() { return reg_; }

   private:
    RegisterPhotoICP reg_;
};

}  // namespace rgbd360
