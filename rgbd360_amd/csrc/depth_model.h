// depth_model.h -- the intrinsic depth-distortion model Frame360::undistort applies to every sensor image before the clouds are built
// (include/Frame360.h:293-311, 1084-1097 -> calib->intrinsic_model_[sensor].undistort, Calib360.h:104-119): CLAMS'
// DiscreteDepthDistortionModel (Teichman et al.; vendored by the reference under OpenNI2_Grabber/third_party/CLAMS, files
// Calibration/Intrinsics/distortion_model1..8).  The image is cut into bins of bin_width x bin_height pixels; every bin holds a
// "frustum" of num_bins depth slices of bin_depth metres with a multiplier each; a measured z becomes z * m, m interpolated linearly
// between the two slices whose centres enclose z when both have seen at least 50 training examples, else the slice's own multiplier.
// File layout (restated from the reference's serialisation code): the line "DiscreteDepthDistortionModel v01", then raw little-endian
// scalars width, height, bin_width, bin_height (int32), bin_depth (float64), num_bins_x, num_bins_y (int32), then per frustum (row of
// bins after row of bins) max_dist (float64), num_bins (int32), bin_depth (float64) and four float32 vectors {counts, total numerators,
// total denominators, multipliers}, each written as {4, rows, cols} (int32) + data.  Host code; no device work (8 x 320 x 240 pixels).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace depthmodel {

struct Frustum {
    double max_dist = 0, bin_depth = 0;
    int num_bins = 0;
    std::vector<float> counts, multipliers;
};
struct Model {
    int width = 0, height = 0, bin_width = 0, bin_height = 0, num_bins_x = 0, num_bins_y = 0;
    double bin_depth = 0;
    std::vector<Frustum> frustums;      // [num_bins_y][num_bins_x]
};

inline bool read_exact(std::FILE* f, void* dst, size_t n) { return std::fread(dst, 1, n, f) == n; }
inline bool read_vec(std::FILE* f, std::vector<float>& v) {
    int32_t hdr[3];
    if (!read_exact(f, hdr, sizeof(hdr)) || hdr[0] != 4 || hdr[1] < 0 || hdr[2] < 0 || (long long)hdr[1] * hdr[2] > (1 << 20)) return false;
    v.resize((size_t)hdr[1] * hdr[2]);
    return v.empty() || read_exact(f, v.data(), v.size() * 4);
}
// 0 ok, 1 cannot open, 2 not a model file / truncated
inline int load(const char* path, Model& M) {
    std::FILE* f = std::fopen(path, "rb");
    if (!f) return 1;
    int rc = 2;
    char line[64];
    static const char kMagic[] = "DiscreteDepthDistortionModel v01\n";
    do {
        if (!std::fgets(line, sizeof(line), f) || std::strcmp(line, kMagic) != 0) break;
        int32_t a[4], b[2];
        if (!read_exact(f, a, sizeof(a)) || !read_exact(f, &M.bin_depth, 8) || !read_exact(f, b, sizeof(b))) break;
        M.width = a[0]; M.height = a[1]; M.bin_width = a[2]; M.bin_height = a[3]; M.num_bins_x = b[0]; M.num_bins_y = b[1];
        if (M.width < 1 || M.height < 1 || M.bin_width < 1 || M.bin_height < 1 || M.num_bins_x < 1 || M.num_bins_y < 1 ||
            (long long)M.num_bins_x * M.num_bins_y > (1 << 22) || !(M.bin_depth > 0))
            break;
        M.frustums.assign((size_t)M.num_bins_x * M.num_bins_y, Frustum());
        bool ok = true;
        for (Frustum& F : M.frustums) {
            std::vector<float> skip;
            int32_t nb;
            ok = read_exact(f, &F.max_dist, 8) && read_exact(f, &nb, 4) && read_exact(f, &F.bin_depth, 8) && read_vec(f, F.counts) && read_vec(f, skip) &&
                 read_vec(f, skip) && read_vec(f, F.multipliers);
            if (!ok) break;
            F.num_bins = nb;
            ok = nb >= 1 && (int)F.counts.size() == nb && (int)F.multipliers.size() == nb && F.bin_depth > 0;
            if (!ok) break;
        }
        if (ok) rc = 0;
    } while (false);
    std::fclose(f);
    return rc;
}
// DiscreteDepthDistortionModel::downsampleParams: the model of the full-resolution sensor used on images `step` times smaller
inline bool downsample(Model& M, int step) {
    if (step < 1 || M.bin_width % step != 0 || M.bin_height % step != 0) return false;
    M.width /= step; M.height /= step; M.bin_width /= step; M.bin_height /= step;
    return true;
}
// DiscreteFrustum::index / undistort / interpolatedUndistort, operation for operation (float z, double interpolation weights)
inline int slice_of(const Frustum& F, float z) {
    const double q = std::floor((double)z / F.bin_depth);
    if (!(q < (double)(F.num_bins - 1))) return F.num_bins - 1;      // beyond the last slice, +Inf (and NaN, which undistort() never passes): no float -> int conversion out of range
    return q < 0 ? -1 : (int)q;
}
inline void undistort_px(const Frustum& F, float* z) {
    const int idx = slice_of(F, *z);
    const float start = (float)(F.bin_depth * idx);
    const int idx1 = ((double)(*z - start) < F.bin_depth / 2) ? idx : idx + 1;
    const int idx0 = idx1 - 1;
    if (idx0 < 0 || idx1 >= F.num_bins || F.counts[(size_t)idx0] < 50 || F.counts[(size_t)idx1] < 50) {
        *z *= F.multipliers[(size_t)(idx < 0 ? 0 : idx)];
        return;
    }
    const double z0 = (idx0 + 1) * F.bin_depth - F.bin_depth * 0.5;
    const double coeff1 = ((double)*z - z0) / F.bin_depth;
    const double coeff0 = 1.0 - coeff1;
    const double mult = coeff0 * (double)F.multipliers[(size_t)idx0] + coeff1 * (double)F.multipliers[(size_t)idx1];
    *z = (float)((double)*z * mult);
}
// DiscreteDepthDistortionModel::undistort on a rows x cols float32 image in metres (0 = no measurement), in place.  The reference
// loops over 240 x 320 whatever the model says; here the image has to be the model's size.
inline bool undistort(const Model& M, float* depth, size_t step_bytes, int rows, int cols) {
    if (rows != M.height || cols != M.width) return false;
    for (int v = 0; v < rows; ++v) {
        float* row = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(depth) + (size_t)v * step_bytes);
        const int yb = std::min(v / M.bin_height, M.num_bins_y - 1);
        for (int u = 0; u < cols; ++u) {
            if (row[u] == 0.f || !(row[u] > 0.f)) continue;      // (no measurement; a negative or NaN depth has no slice either)
            undistort_px(M.frustums[(size_t)yb * M.num_bins_x + (size_t)std::min(u / M.bin_width, M.num_bins_x - 1)], &row[u]);
        }
    }
    return true;
}

}  // namespace depthmodel
