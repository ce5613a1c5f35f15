// rgbd360_host.cpp -- the host-only entry points of the C ABI (include/rgbd360_hip.h): the .bin frame reader (Frame360::loadFrame), the
// sensors' intrinsic depth models (Frame360::undistort), PbMap plane registration and the plane pooling steps of Frame360::getPlanes.
// No device code, no context: a translation unit of its own since round 6 (compiled in a second).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <vector>

#include "../../include/rgbd360_hip.h"
#include "depth_model.h"
#include "pbmap_register.h"

// ---- the sensors' intrinsic depth model (depth_model.h): host only ----
struct rgbd360_depth_model {
    depthmodel::Model m;
};
extern "C" int rgbd360_depth_model_load(const char* path, int downsample, rgbd360_depth_model** out) {
    if (!path || !out || downsample < 1) return -1;
    *out = nullptr;
    try {
        std::unique_ptr<rgbd360_depth_model> M(new rgbd360_depth_model());
        const int rc = depthmodel::load(path, M->m);
        if (rc) return rc;
        if (!depthmodel::downsample(M->m, downsample)) return 2;
        *out = M.release();
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}
extern "C" void rgbd360_depth_model_free(rgbd360_depth_model* model) { delete model; }
extern "C" int rgbd360_depth_model_info(const rgbd360_depth_model* model, int dims[6], double* bin_depth) {
    if (!model || !dims) return -1;
    const depthmodel::Model& m = model->m;
    dims[0] = m.width; dims[1] = m.height; dims[2] = m.bin_width; dims[3] = m.bin_height; dims[4] = m.num_bins_x; dims[5] = m.num_bins_y;
    if (bin_depth) *bin_depth = m.bin_depth;
    return 0;
}
extern "C" int rgbd360_depth_model_undistort(const rgbd360_depth_model* model, float* depth_m, size_t depth_step, int rows, int cols) {
    if (!model || !depth_m || rows < 1 || cols < 1 || depth_step < (size_t)cols * 4) return -1;
    return depthmodel::undistort(model->m, depth_m, depth_step, rows, cols) ? 0 : -1;
}


// ---------------------------------------------------------------------------------------------------------
// Frame360 input side: the .bin reader and the spherical stitcher, i.e. the two steps between
// the sensor rig's raw frames and rgbd360_set_target / _source.
// ---------------------------------------------------------------------------------------------------------
extern "C" int rgbd360_load_frame_bin(const char* path, uint8_t* rgb_out, uint16_t* depth_out, int* rows, int* cols) {
    // Frame360::loadFrame (Frame360.h:231-266): boost::archive::binary_iarchive of 8 x {cv::Mat rgb 8UC3, cv::Mat depth
    // 16UC1} + a timestamp Mat; every Mat = int32 cols, int32 rows, uint64 elemSize, uint64 cvType, raw bytes
    // (cvmat_serialization.h:23-36) behind the archive's 45-byte header.
    if (!path || !rows || !cols) return -1;
    // with output buffers the caller states the size they were allocated for (the values of the size query); a file whose
    // records do not have exactly that size is refused instead of being read into them
    const bool have_buffers = rgb_out || depth_out;
    const int want_rows = *rows, want_cols = *cols;
    if (have_buffers && (want_rows <= 0 || want_cols <= 0)) return -1;
    FILE* f = fopen(path, "rb");
    if (!f) return -2;
    int rc = 0;
    if (fseek(f, 45, SEEK_SET) != 0) rc = -3;
    for (int m = 0; m < 16 && rc == 0; ++m) {
        int32_t c = 0, r = 0;
        uint64_t elem = 0, type = 0;
        if (fread(&c, 4, 1, f) != 1 || fread(&r, 4, 1, f) != 1 || fread(&elem, 8, 1, f) != 1 || fread(&type, 8, 1, f) != 1) { rc = -3; break; }
        const bool is_rgb = (m % 2) == 0;
        if (c <= 0 || r <= 0 || c > 8192 || r > 8192 || elem != (is_rgb ? 3u : 2u) || type != (is_rgb ? 16u : 2u)) { rc = -4; break; }   // CV_8UC3 = 16, CV_16UC1 = 2
        if (have_buffers && (r != want_rows || c != want_cols)) { rc = -4; break; }
        if (m == 0) { *rows = r; *cols = c; }
        else if (r != *rows || c != *cols) { rc = -4; break; }
        const size_t bytes = (size_t)c * r * elem;
        void* dst = is_rgb ? (void*)(rgb_out ? rgb_out + (size_t)(m / 2) * bytes : nullptr)
                           : (void*)(depth_out ? depth_out + (size_t)(m / 2) * (bytes / 2) : nullptr);
        if (dst) {
            if (fread(dst, 1, bytes, f) != bytes) rc = -3;
        } else if (fseek(f, (long)bytes, SEEK_CUR) != 0) rc = -3;
    }
    fclose(f);
    return rc;
}

// ---- PbMap plane registration (host only) ---------------------------------------------------
extern "C" void rgbd360_pbmap_default_params(rgbd360_pbmap_params* p, int odometry) {
    if (p) pbm::default_params(p, odometry);
}

extern "C" int rgbd360_register_planes(const rgbd360_plane* ref, int n_ref, const rgbd360_plane* trg, int n_trg, int max_match_planes,
                                       int regist_mode, const rgbd360_pbmap_params* params, float pose_out[16], float info_out[36],
                                       int32_t* match_out, int* n_matched_out, float* area_matched_out) {
    rgbd360_pbmap_params def;
    if (!params) {
        pbm::default_params(&def, (regist_mode == 2 || regist_mode == 3) ? 1 : 0);
        params = &def;
    }
    try {
        return pbm::register_planes(ref, n_ref, trg, n_trg, max_match_planes, regist_mode, params, pose_out, info_out, match_out,
                                    n_matched_out, area_matched_out);
    } catch (const std::exception&) {       // allocation failure: nothing may cross the C boundary
        return -1;
    }
}

extern "C" int rgbd360_merge_planes(const rgbd360_plane* planes, int n, float max_curvature, float min_area, float max_elongation, float cos_normal,
                                    float dist_d, float proximity, float normal_offset, rgbd360_plane* out, int max_out, int* n_out) {
    if (n < 0 || (n > 0 && !planes) || !out || !n_out || max_out < 0) return -1;
    try {
        const pbm::MergeParams M{max_curvature, cos_normal, dist_d, proximity, normal_offset, min_area, max_elongation};
        const std::vector<rgbd360_plane> v = pbm::merge_planes(planes, n, M);
        *n_out = (int)v.size();
        if ((int)v.size() > max_out) return -1;
        for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}

extern "C" int rgbd360_group_planes(const rgbd360_plane* planes, const int* n_per_sensor, int n_sensors, float max_curvature, float min_area,
                                    float cos_normal, float dist_d, float max_dist_hull, float max_dist_parallel_hull, rgbd360_plane* out,
                                    int max_out, int* n_out) {
    if (n_sensors < 1 || !n_per_sensor || !out || !n_out || max_out < 0) return -1;
    long long n = 0;
    for (int s = 0; s < n_sensors; ++s) {
        if (n_per_sensor[s] < 0) return -1;
        n += n_per_sensor[s];
    }
    if (n > 0 && !planes) return -1;
    try {
        pbm::GroupParams G;
        G.max_curvature = max_curvature; G.min_area = min_area; G.cos_normal = cos_normal; G.dist_d = dist_d;
        G.max_dist_hull = max_dist_hull; G.max_dist_parallel_hull = max_dist_parallel_hull;
        const std::vector<rgbd360_plane> v = pbm::group_planes(planes, n_per_sensor, n_sensors, G);
        *n_out = (int)v.size();
        if ((int)v.size() > max_out) return -1;
        for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}

extern "C" int rgbd360_pool_sensor_planes(const rgbd360_plane* planes, int n, float max_curvature, float min_area, float max_elongation, float cos_normal,
                                         float dist_normal, float proximity, rgbd360_plane* out, int max_out, int* n_out) {
    if (n < 0 || (n > 0 && !planes) || !out || !n_out || max_out < 0) return -1;
    try {
        pbm::SensorPoolParams P;
        P.max_curvature = max_curvature; P.min_area = min_area; P.max_elongation = max_elongation;
        P.cos_normal = cos_normal; P.dist_normal = dist_normal; P.proximity = proximity;
        const std::vector<rgbd360_plane> v = pbm::pool_sensor_planes(planes, n, P);
        *n_out = (int)v.size();
        if ((int)v.size() > max_out) return -1;
        for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}

