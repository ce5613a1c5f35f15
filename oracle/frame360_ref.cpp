// =====================================================================================
//  oracle/frame360_ref.cpp  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
//  CPU restatement of the two PCL algorithms Frame360 runs on its organised cloud before handing planes to
//  mrpt::pbmap:
//    Frame360.h:949-977, Frame360_stereo.h:854-882
//      pcl::IntegralImageNormalEstimation  AVERAGE_3D_GRADIENT, maxDepthChangeFactor 0.02 / 0.05,
//                                          normalSmoothingSize 8, depth-dependent smoothing
//      pcl::OrganizedMultiPlaneSegmentation minInliers 80 / 40, angular 0.0398 / 0.05 rad, distance 0.02 / 0.05
//
//  PARITY UNPINNED / THIRD-PARTY: PCL (>= 1.7, version not pinned by the reference's CMakeLists.txt:19) is not in the
//  reference tree and not installed; the algorithms are restated from PCL 1.7's published sources
//  (features/integral_image_normal.hpp computeFeature / computeFeatureFull / computePointNormal,
//   segmentation/organized_multi_plane_segmentation.hpp segment, plane_coefficient_comparator.h compare,
//   organized_connected_component_segmentation.hpp).  Deliberate differences, shared with the HIP path:
//    * window sums are taken directly in double (PCL: double integral images) -- same values up to rounding;
//    * region moments are accumulated in double (PCL 1.7 uses float accumulators);
//    * segmentAndRefine's boundary refinement step and everything mrpt::pbmap does afterwards (hulls, areas,
//      colours, merging; Frame360.h:984-1075) are not part of this stage;
//    * depth_mode 1 replaces PCL's use of the z coordinate as "depth" by the range |p|, which is what makes the
//      depth-dependent thresholds meaningful on a full sphere (depth_mode 0 is PCL-faithful);
//    * the chamfer distance map ignores PCL's row-wrap reads in the first/last column (those columns lie in the
//      border band that is set to NaN anyway).
// =====================================================================================
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <vector>

namespace {

inline bool finite3(const float* p) { return std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]); }
inline float depth_of(const float* p, int depth_mode) {
    return depth_mode == 0 ? p[2] : sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
}

// integral_image_normal.hpp computeFeature: depth-change map + two-pass chamfer (1 / 1.4) distance map
void distance_map(const float* xyz, int rows, int cols, float max_depth_change_factor, int depth_mode, std::vector<float>& dist) {
    const size_t n = (size_t)rows * cols;
    std::vector<unsigned char> change(n, 255);
    for (int ri = 0; ri < rows - 1; ++ri)
        for (int ci = 0; ci < cols - 1; ++ci) {
            const size_t index = (size_t)ri * cols + ci;
            const float depth = depth_of(xyz + 3 * index, depth_mode);
            const float depthR = depth_of(xyz + 3 * (index + 1), depth_mode);
            const float depthD = depth_of(xyz + 3 * (index + cols), depth_mode);
            const float ddc = max_depth_change_factor * (fabsf(depth) + 1.0f) * 2.0f;
            if (fabs(depth - depthR) > ddc || !std::isfinite(depth) || !std::isfinite(depthR)) {
                change[index] = 0;
                change[index + 1] = 0;
            }
            if (fabs(depth - depthD) > ddc || !std::isfinite(depth) || !std::isfinite(depthD)) {
                change[index] = 0;
                change[index + cols] = 0;
            }
        }
    dist.resize(n);
    for (size_t i = 0; i < n; ++i) dist[i] = change[i] == 0 ? 0.0f : (float)(cols + rows);
    for (int ri = 1; ri < rows; ++ri)
        for (int ci = 1; ci < cols - 1; ++ci) {
            float* cur = &dist[(size_t)ri * cols];
            const float* prev = cur - cols;
            const float upLeft = prev[ci - 1] + 1.4f, up = prev[ci] + 1.0f, upRight = prev[ci + 1] + 1.4f, left = cur[ci - 1] + 1.0f;
            const float minValue = std::min(std::min(upLeft, up), std::min(left, upRight));
            if (minValue < cur[ci]) cur[ci] = minValue;
        }
    for (int ri = rows - 2; ri >= 0; --ri)
        for (int ci = cols - 2; ci >= 1; --ci) {
            float* cur = &dist[(size_t)ri * cols];
            const float* next = cur + cols;
            const float lowerLeft = next[ci - 1] + 1.4f, lower = next[ci] + 1.0f, lowerRight = next[ci + 1] + 1.4f, right = cur[ci + 1] + 1.0f;
            const float minValue = std::min(std::min(lowerLeft, lower), std::min(right, lowerRight));
            if (minValue < cur[ci]) cur[ci] = minValue;
        }
}

}  // namespace

extern "C" {

// Truncated chamfer distance map exactly as above (exposed so the device's map can be compared on its own).
void oracle_f360_distance_map(const float* xyz, int rows, int cols, float max_depth_change_factor, int depth_mode, float* out) {
    std::vector<float> d;
    distance_map(xyz, rows, cols, max_depth_change_factor, depth_mode, d);
    memcpy(out, d.data(), d.size() * sizeof(float));
}

// IntegralImageNormalEstimation, AVERAGE_3D_GRADIENT, depth-dependent smoothing, BORDER_POLICY_IGNORE.
// normals: n x 3, NaN where PCL leaves the normal undefined.  window_out (optional): the rect size used, 0 if none.
void oracle_f360_normals(const float* xyz, int rows, int cols, float max_depth_change_factor, float normal_smoothing_size,
                         int depth_mode, float* normals, int* window_out) {
    const size_t n = (size_t)rows * cols;
    const float qnan = std::numeric_limits<float>::quiet_NaN();
    for (size_t i = 0; i < 3 * n; ++i) normals[i] = qnan;
    if (window_out) memset(window_out, 0, n * sizeof(int));
    std::vector<float> dist;
    distance_map(xyz, rows, cols, max_depth_change_factor, depth_mode, dist);
    // initAverage3DGradientMethod: central differences, zero on the image border
    std::vector<float> dx(3 * n, 0.f), dy(3 * n, 0.f);
    for (int r = 1; r < rows - 1; ++r)
        for (int c = 1; c < cols - 1; ++c) {
            const size_t i = (size_t)r * cols + c;
            for (int k = 0; k < 3; ++k) {
                dx[3 * i + k] = xyz[3 * (i + 1) + k] - xyz[3 * (i - 1) + k];
                dy[3 * i + k] = xyz[3 * (i + cols) + k] - xyz[3 * (i - cols) + k];
            }
        }
    const int border = (int)normal_smoothing_size;
#pragma omp parallel for
    for (int ri = border; ri < rows - border; ++ri)
        for (int ci = border; ci < cols - border; ++ci) {
            const size_t index = (size_t)ri * cols + ci;
            const float depth = depth_of(xyz + 3 * index, depth_mode);
            if (!std::isfinite(depth)) continue;
            const float smoothing = std::min(dist[index], normal_smoothing_size + depth / 10.0f);
            if (!(smoothing > 2.0f)) continue;
            const int rect = (int)smoothing;                 // setRectSize(rect, rect)
            const int x0 = ci - rect / 2, y0 = ri - rect / 2;
            double gx[3] = {0, 0, 0}, gy[3] = {0, 0, 0};
            unsigned count_x = 0, count_y = 0;
            for (int y = y0; y < y0 + rect; ++y)
                for (int x = x0; x < x0 + rect; ++x) {
                    if (x < 0 || y < 0 || x >= cols || y >= rows) continue;
                    const size_t j = (size_t)y * cols + x;
                    if (finite3(&dx[3 * j])) {
                        ++count_x;
                        for (int k = 0; k < 3; ++k) gx[k] += dx[3 * j + k];
                    }
                    if (finite3(&dy[3 * j])) {
                        ++count_y;
                        for (int k = 0; k < 3; ++k) gy[k] += dy[3 * j + k];
                    }
                }
            if (window_out) window_out[index] = rect;
            if (count_x == 0 || count_y == 0) continue;
            double nv[3] = {gy[1] * gx[2] - gy[2] * gx[1], gy[2] * gx[0] - gy[0] * gx[2], gy[0] * gx[1] - gy[1] * gx[0]};   // gradient_y x gradient_x
            const double len2 = nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2];
            if (len2 == 0.0) continue;
            const double inv = 1.0 / sqrt(len2);
            float nx = (float)(nv[0] * inv), ny = (float)(nv[1] * inv), nz = (float)(nv[2] * inv);
            // flipNormalTowardsViewpoint, viewpoint (0,0,0)
            const float* p = xyz + 3 * index;
            if ((-p[0]) * nx + (-p[1]) * ny + (-p[2]) * nz < 0) {
                nx = -nx; ny = -ny; nz = -nz;
            }
            normals[3 * index] = nx; normals[3 * index + 1] = ny; normals[3 * index + 2] = nz;
        }
}

struct oracle_plane {
    float centroid[3];
    float normal[3];
    float d;            // n . x + d = 0
    float curvature;    // lambda_min / trace(cov)
    int   count;
    int   root;         // smallest pixel index of the region (PCL's label order)
    float area;         // 12 sqrt(l1 l2), l1 <= l2 the in-plane eigenvalues of the inlier covariance (rgbd360_hip.h)
    float elongation;   // sqrt(l2 / l1)
    float ppal_dir[3];  // eigenvector of l2
};

// smallest eigenpair of a symmetric 3x3 (cyclic Jacobi, double)
static void smallest_eigen(const double C[3][3], double& eval, double evec[3], double* others = nullptr) {
    double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    memcpy(A, C, sizeof(A));
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2 * A[p][q]);
                const double t = (theta >= 0 ? 1 : -1) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int m = 0;
    for (int k = 1; k < 3; ++k)
        if (A[k][k] < A[m][m]) m = k;
    eval = A[m][m];
    for (int k = 0; k < 3; ++k) evec[k] = V[k][m];
    if (others) {       // the two in-plane eigenpairs, ascending: others[0..1] = eigenvalues, others[2..4] = eigenvector of the larger
        const int a = (m + 1) % 3, b = (m + 2) % 3;
        const int lo = A[a][a] <= A[b][b] ? a : b, hi = lo == a ? b : a;
        others[0] = A[lo][lo];
        others[1] = A[hi][hi];
        for (int k = 0; k < 3; ++k) others[2 + k] = V[k][hi];
    }
}

// OrganizedMultiPlaneSegmentation::segment.  labels: n int32 (root pixel index of the region, -1 for non-finite
// points).  Returns the number of planes written (<= max_planes), in PCL's order (by first pixel).
int oracle_f360_plane_segment(const float* xyz, const float* normals, int rows, int cols, int min_inliers, float angular_threshold,
                              float distance_threshold, float max_curvature, int depth_mode, int* labels, oracle_plane* planes,
                              int max_planes) {
    const size_t n = (size_t)rows * cols;
    std::vector<float> plane_d(n);
    for (size_t i = 0; i < n; ++i)
        plane_d[i] = xyz[3 * i] * normals[3 * i] + xyz[3 * i + 1] * normals[3 * i + 1] + xyz[3 * i + 2] * normals[3 * i + 2];
    const float cos_thr = cosf(angular_threshold);
    auto compare = [&](size_t a, size_t b) {          // PlaneCoefficientComparator::compare, depth-dependent threshold
        const float z = depth_of(xyz + 3 * a, depth_mode);
        const float thr = distance_threshold * z * z;
        const float dot = normals[3 * a] * normals[3 * b] + normals[3 * a + 1] * normals[3 * b + 1] + normals[3 * a + 2] * normals[3 * b + 2];
        return (fabs(plane_d[a] - plane_d[b]) < thr) && (dot > cos_thr);
    };
    std::vector<int> parent(n);
    auto find = [&](int x) {
        while (parent[x] != x) {
            parent[x] = parent[parent[x]];
            x = parent[x];
        }
        return x;
    };
    for (size_t i = 0; i < n; ++i) parent[i] = finite3(xyz + 3 * i) ? (int)i : -1;
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            const size_t i = (size_t)r * cols + c;
            if (parent[i] < 0) continue;
            if (c > 0 && parent[i - 1] >= 0 && compare(i, i - 1)) {
                int a = find((int)i), b = find((int)(i - 1));
                if (a != b) parent[std::max(a, b)] = std::min(a, b);
            }
            if (r > 0 && parent[i - cols] >= 0 && compare(i, i - cols)) {
                int a = find((int)i), b = find((int)(i - cols));
                if (a != b) parent[std::max(a, b)] = std::min(a, b);
            }
        }
    std::vector<int> count(n, 0);
    for (size_t i = 0; i < n; ++i) {
        labels[i] = parent[i] < 0 ? -1 : find((int)i);
        if (labels[i] >= 0) ++count[labels[i]];
    }
    int n_planes = 0;
    std::vector<int> roots;
    for (size_t i = 0; i < n; ++i)
        if (labels[i] == (int)i && count[i] > min_inliers) roots.push_back((int)i);
    std::vector<int> slot(n, -1);
    for (size_t k = 0; k < roots.size(); ++k) slot[roots[k]] = (int)k;
    std::vector<double> mom(roots.size() * 9, 0.0);
    for (size_t i = 0; i < n; ++i) {
        if (labels[i] < 0 || slot[labels[i]] < 0) continue;
        double* m = &mom[(size_t)slot[labels[i]] * 9];
        const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        m[0] += x; m[1] += y; m[2] += z;
        m[3] += x * x; m[4] += x * y; m[5] += x * z; m[6] += y * y; m[7] += y * z; m[8] += z * z;
    }
    for (size_t k = 0; k < roots.size() && n_planes < max_planes; ++k) {
        const double* m = &mom[k * 9];
        const double N = count[roots[k]];
        const double cx = m[0] / N, cy = m[1] / N, cz = m[2] / N;
        const double C[3][3] = {{m[3] / N - cx * cx, m[4] / N - cx * cy, m[5] / N - cx * cz},
                                {m[4] / N - cx * cy, m[6] / N - cy * cy, m[7] / N - cy * cz},
                                {m[5] / N - cx * cz, m[7] / N - cy * cz, m[8] / N - cz * cz}};
        double ev, v[3], inplane[5];
        smallest_eigen(C, ev, v, inplane);
        double d = -(v[0] * cx + v[1] * cy + v[2] * cz);
        // orient towards the viewpoint (origin): vp - centroid
        if ((-cx) * v[0] + (-cy) * v[1] + (-cz) * v[2] < 0) {
            v[0] = -v[0]; v[1] = -v[1]; v[2] = -v[2];
            d = -(v[0] * cx + v[1] * cy + v[2] * cz);
        }
        const double tr = C[0][0] + C[1][1] + C[2][2];
        const double curvature = tr != 0 ? fabs(ev / tr) : 0;
        if (!(curvature < max_curvature)) continue;
        oracle_plane& P = planes[n_planes++];
        P.centroid[0] = (float)cx; P.centroid[1] = (float)cy; P.centroid[2] = (float)cz;
        P.normal[0] = (float)v[0]; P.normal[1] = (float)v[1]; P.normal[2] = (float)v[2];
        P.d = (float)d;
        P.curvature = (float)curvature;
        P.count = count[roots[k]];
        P.root = roots[k];
        const double l1 = std::max(inplane[0], 0.0), l2 = std::max(inplane[1], 0.0);
        P.area = (float)(12.0 * sqrt(l1 * l2));
        P.elongation = (float)(l1 > 0 ? sqrt(l2 / l1) : INFINITY);
        for (int k2 = 0; k2 < 3; ++k2) P.ppal_dir[k2] = (float)inplane[2 + k2];
    }
    return n_planes;
}


// ------------------------------------------------------------------------------------
// Colour descriptors of the planar regions: the roles of mrpt::pbmap::Plane::calcMainColor2 / calcPlaneHistH, which
// Frame360.h:1045-1046 and Frame360_stereo.h:949-950 call for every plane, and whose results the PbMap matcher's unary colour
// constraint reads (configLocaliser_spherical.ini:19-21).  THIRD-PARTY (MRPT 1.x, src/Plane.cpp), not in the reference tree, version
// not pinned: PARITY UNPINNED.  Restated from the published method:
//   * per inlier the normalised colour (R, G, B) / (R + G + B) (pixels with R + G + B = 0 have none), its mean and standard
//     deviation over the region (calcMainColor's estimate; calcMainColor2 refines it to the mean-shift mode of a 2000-pixel
//     subsample -- the same value for a region of one colour, not restated) and the mean intensity R + G + B;
//   * per inlier the hue of the RGB -> HSV conversion in 72 bins of 5 degrees when the pixel is saturated (V > 0.2 and S > 0.2),
//     else a "dark" (V <= 0.2) or "unsaturated" bin: the 74-bin layout of hist_H, normalised by the inlier count.
// Integer arithmetic throughout (the device does the same: sums are order independent and comparable bit for bit):
//   q_c = floor(C * 65536 / (R + G + B));  sums of q_c, q_c^2, R + G + B, pixel counts;
//   bin = floor((24 k delta + 12 (x - y)) / delta) mod 72 with delta = max - min, (k, x, y) = (0, G, B) when R is the maximum,
//   (1, B, R) when G is (and R is not), (2, R, G) otherwise.
// labels: per cloud pixel the region's root pixel index (-1: none); roots: the regions wanted; cloud pixel (r, c) takes image pixel
// (r step + step / 2, c step + step / 2) (DownsampleRGBD.h:240, 285-287).  out: n_planes x 82 uint64
// {sum qR, qG, qB, sum qR^2, qG^2, qB^2, sum (R + G + B), pixels with colour, 74 bins}.
void oracle_f360_plane_colour(const int* labels, int rows, int cols, const uint8_t* rgb, size_t rgb_step, int step, const int* roots,
                              int n_planes, uint64_t* out) {
    const size_t n = (size_t)rows * cols;
    std::vector<int> plane_of(n, -1);
    for (int k = 0; k < n_planes; ++k) plane_of[roots[k]] = k;
    std::fill(out, out + (size_t)n_planes * 82, (uint64_t)0);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            const int l = labels[(size_t)r * cols + c];
            if (l < 0 || plane_of[l] < 0) continue;
            uint64_t* o = out + (size_t)plane_of[l] * 82;
            const uint8_t* px = rgb + (size_t)(r * step + step / 2) * rgb_step + 3 * (size_t)(c * step + step / 2);
            const unsigned R = px[0], G = px[1], B = px[2], S = R + G + B;
            if (S) {
                const uint64_t q[3] = {(uint64_t)(R << 16) / S, (uint64_t)(G << 16) / S, (uint64_t)(B << 16) / S};
                for (int k = 0; k < 3; ++k) {
                    o[k] += q[k];
                    o[3 + k] += q[k] * q[k];
                }
                o[6] += S;
                o[7] += 1;
            }
            const unsigned mx = std::max(R, std::max(G, B)), mn = std::min(R, std::min(G, B)), delta = mx - mn;
            int bin;
            if (mx * 5u <= 255u) bin = 72;
            else if (delta * 5u <= mx) bin = 73;
            else {
                int num;
                if (mx == R) num = 12 * ((int)G - (int)B);
                else if (mx == G) num = 24 * (int)delta + 12 * ((int)B - (int)R);
                else num = 48 * (int)delta + 12 * ((int)R - (int)G);
                if (num < 0) num += 72 * (int)delta;
                bin = num / (int)delta;
                if (bin >= 72) bin -= 72;
            }
            o[8 + bin] += 1;
        }
}

// ------------------------------------------------------------------------------------
// The `refine` half of pcl::OrganizedMultiPlaneSegmentation::segmentAndRefine (Frame360.h:977, :868; Frame360_stereo.h:882).
// THIRD-PARTY (PCL >= 1.7, segmentation/organized_multi_plane_segmentation.hpp `refine` + plane_refinement_comparator.h), restated
// from the published source, parity unpinned.  After `segment`, the regions that became planes ("refine labels") grow into
// neighbouring pixels of regions that did not: two raster passes with in-place label updates,
//   pass 1, rows 0 .. H-2 top-down, columns 0 .. W-2 left to right: current -> right neighbour, then current -> lower neighbour;
//   pass 2, rows H-1 .. 1 bottom-up, columns W-1 .. 0 right to left: current -> left neighbour, then current -> upper neighbour;
// a neighbour takes the current pixel's label when the current label is a plane's, the neighbour's is not, and the neighbour's
// point lies within the comparator's distance threshold of that plane (PlaneRefinementComparator::compare: |a x + b y + c z + d|
// in float against 0.02 m -- the default-constructed refinement comparator is never given mps.setDistanceThreshold's value and is
// not depth dependent).  An invalid current OR first-neighbour label `continue`s the loop body, which also skips the second check
// (kept).  One deviation: in pass 2 PCL reads labels[current_row + colIdx - 1] at colIdx == 0, i.e. the LAST pixel of the row
// above, as "left neighbour" (an out-of-row access); here column 0 has no left neighbour and goes straight to the upper check.
// The planes keep centroid / normal / d / curvature of `segment` (PCL builds the PlanarRegion from the pre-refinement centroid and
// covariance); their inlier sets grow, so count and the extent descriptors of the inliers (area, elongation, ppal_dir) are
// recomputed -- what Frame360.h:1010-1037 derives from the refined inlier cloud.
// labels: in/out (root index per pixel, -1 invalid); planes: in/out.  Returns the number of pixels relabelled.
int oracle_f360_plane_refine(const float* xyz, int rows, int cols, int* labels, oracle_plane* planes, int n_planes, float distance_threshold) {
    const size_t n = (size_t)rows * cols;
    std::vector<int> model_of(n, -1);          // label (root pixel) -> plane index; >= 0 <=> refine label
    for (int k = 0; k < n_planes; ++k) model_of[planes[k].root] = k;
    const std::vector<int> before(labels, labels + n);
    auto compare = [&](size_t idx1, size_t idx2) {
        const int current_label = labels[idx1], next_label = labels[idx2];
        if (!(model_of[current_label] >= 0 && !(model_of[next_label] >= 0))) return false;
        const oracle_plane& m = planes[model_of[current_label]];
        const float* pt = xyz + 3 * idx2;
        const double ptp_dist = fabs(m.normal[0] * pt[0] + m.normal[1] * pt[1] + m.normal[2] * pt[2] + m.d);
        return ptp_dist < distance_threshold;
    };
    // first pass: top to bottom, left to right
    for (int r = 0; r < rows - 1; ++r)
        for (int c = 0; c < cols - 1; ++c) {
            const size_t cur = (size_t)r * cols + c;
            const int current_label = labels[cur], right_label = labels[cur + 1];
            if (current_label < 0 || right_label < 0) continue;
            if (compare(cur, cur + 1)) labels[cur + 1] = current_label;
            const int lower_label = labels[cur + cols];
            if (lower_label < 0) continue;
            if (compare(cur, cur + cols)) labels[cur + cols] = current_label;
        }
    // second pass: bottom to top, right to left
    for (int r = rows - 1; r >= 1; --r)
        for (int c = cols - 1; c >= 0; --c) {
            const size_t cur = (size_t)r * cols + c;
            const int current_label = labels[cur];
            if (c >= 1) {
                const int left_label = labels[cur - 1];
                if (current_label < 0 || left_label < 0) continue;
                if (compare(cur, cur - 1)) labels[cur - 1] = current_label;
            } else if (current_label < 0) {
                continue;
            }
            const int upper_label = labels[cur - cols];
            if (upper_label < 0) continue;
            if (compare(cur, cur - cols)) labels[cur - cols] = current_label;
        }
    // grown inlier sets: count and extent descriptors
    int changed = 0;
    std::vector<double> mom((size_t)n_planes * 9, 0.0);
    std::vector<int> count(n_planes, 0);
    for (size_t i = 0; i < n; ++i) {
        if (labels[i] != before[i]) ++changed;
        if (labels[i] < 0 || model_of[labels[i]] < 0) continue;
        const int k = model_of[labels[i]];
        double* m = &mom[(size_t)k * 9];
        const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        m[0] += x; m[1] += y; m[2] += z;
        m[3] += x * x; m[4] += x * y; m[5] += x * z; m[6] += y * y; m[7] += y * z; m[8] += z * z;
        ++count[k];
    }
    for (int k = 0; k < n_planes; ++k) {
        const double* m = &mom[(size_t)k * 9];
        const double N = count[k];
        const double cx = m[0] / N, cy = m[1] / N, cz = m[2] / N;
        const double C[3][3] = {{m[3] / N - cx * cx, m[4] / N - cx * cy, m[5] / N - cx * cz},
                                {m[4] / N - cx * cy, m[6] / N - cy * cy, m[7] / N - cy * cz},
                                {m[5] / N - cx * cz, m[7] / N - cy * cz, m[8] / N - cz * cz}};
        double ev, v[3], inplane[5];
        smallest_eigen(C, ev, v, inplane);
        oracle_plane& P = planes[k];
        P.count = count[k];
        const double l1 = std::max(inplane[0], 0.0), l2 = std::max(inplane[1], 0.0);
        P.area = (float)(12.0 * sqrt(l1 * l2));
        P.elongation = (float)(l1 > 0 ? sqrt(l2 / l1) : INFINITY);
        for (int k2 = 0; k2 < 3; ++k2) P.ppal_dir[k2] = (float)inplane[2 + k2];
    }
    return changed;
}


// ------------------------------------------------------------------------------------
// One sensor's organised cloud as Frame360::buildSphereCloud_rgbd360 feeds it to the plane extraction (Frame360.h:479-481):
// CloudRGBD::getPointCloud (OpenNI2_Grabber/FrameRGBD/CloudRGBD.h:107-166: focal 525 * W / 640, centre (W/2 - 0.5, H/2 - 0.5),
// x = (c - ox) * z * inv_fx, y = (r - oy) * z * inv_fy, z = 0.001 * depth) followed by DownsampleRGBD::downsamplePointCloud
// (OpenNI2_Grabber/FrameRGBD/DownsampleRGBD.h:209-300: per step x step block and per coordinate, element n/2 of the sorted
// valid values; NaN when the block has none).  Validity = DownsampleRGBD's test min_depth < z < max_depth on a measured depth
// (> 0); the millimetre / metre mix-up of CloudRGBD.h:133-150 (z in metres compared with thresholds scaled by 1000, which
// no pixel passes) is not reproduced.  out: (rows / step) x (cols / step) x 3.
void oracle_sensor_cloud(const uint16_t* depth, size_t depth_step_bytes, int rows, int cols, int step, float min_depth, float max_depth,
                         float* out) {
    const float res_factor_VGA = cols / 640.0;
    const float focal_length = 525 * res_factor_VGA;
    const float inv_fx = 1.f / focal_length, inv_fy = 1.f / focal_length;
    const float ox = cols / 2 - 0.5, oy = rows / 2 - 0.5;
    const int orows = rows / step, ocols = cols / step;
    const float qnan = std::numeric_limits<float>::quiet_NaN();
    for (int r = 0; r < orows; ++r)
        for (int c = 0; c < ocols; ++c) {
            float xs[16], ys[16], zs[16];
            int n = 0;
            for (int r2 = r * step; r2 < (r + 1) * step; ++r2)
                for (int c2 = c * step; c2 < (c + 1) * step; ++c2) {
                    const uint16_t d = *reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(depth) + (size_t)r2 * depth_step_bytes + 2 * (size_t)c2);
                    const float z = 0.001 * d;             // double product rounded to float, CloudRGBD.h:147
                    if (d > 0 && min_depth < z && z < max_depth) {
                        xs[n] = (c2 - ox) * z * inv_fx;
                        ys[n] = (r2 - oy) * z * inv_fy;
                        zs[n] = z;
                        ++n;
                    }
                }
            float* o = out + 3 * ((size_t)r * ocols + c);
            if (n == 0) {
                o[0] = o[1] = o[2] = qnan;
                continue;
            }
            std::sort(xs, xs + n);
            std::sort(ys, ys + n);
            std::sort(zs, zs + n);
            o[0] = xs[n / 2];
            o[1] = ys[n / 2];
            o[2] = zs[n / 2];
        }
}

// ------------------------------------------------------------------------------------
// pcl::FastBilateralFilter<PointXYZRGBA>::applyFilter with setSigmaS(sigma_s) / setSigmaR(sigma_r), the smoothing Frame360
// applies to every sensor cloud before the planes are segmented (Frame360.h:40, 493-499: sigma_s 10 px, sigma_r 0.05 m).
// THIRD-PARTY (PCL >= 1.7 filters/impl/fast_bilateral.hpp, not in the reference tree, unpinned): restated from the published
// bilateral-grid algorithm (Paris & Durand) as PCL implements it -- only z is filtered; non-finite z enter the grid as the
// largest finite z; grid of (cols-1)/sigma_s + 1 + 4 by (rows-1)/sigma_s + 1 + 4 by (zmax-zmin)/sigma_r + 1 + 4 cells of
// {sum z, count}; two [1 2 1]/4 passes per axis over the interior cells, ping-ponging two arrays whose border cells are never
// written (PCL's swap); z = trilinear(sum z) / trilinear(count).  One deliberate difference, shared with the device: the
// cell sums are accumulated in 2^-20 m fixed point (integers), so they do not depend on the order of the points.
struct BilateralGrid {
    int nx, ny, nz;
    std::vector<float> v;            // {sum z, count} per cell, index ((x * ny) + y) * nz + z
    size_t idx(int x, int y, int z) const { return (((size_t)x * ny) + y) * nz + z; }
};
constexpr double kBilatFixed = 1048576.0;      // 2^20 units per metre

void oracle_fast_bilateral(const float* xyz, int rows, int cols, float sigma_s, float sigma_r, float* out) {
    const size_t n = (size_t)rows * cols;
    memcpy(out, xyz, n * 3 * sizeof(float));
    float base_max = -std::numeric_limits<float>::max(), base_min = std::numeric_limits<float>::max();
    bool found = false;
    for (size_t i = 0; i < n; ++i) {
        const float z = xyz[3 * i + 2];
        if (std::isfinite(z)) {
            base_max = std::max(base_max, z);
            base_min = std::min(base_min, z);
            found = true;
        }
    }
    if (!found) return;
    const float base_delta = base_max - base_min;
    const int pad_xy = 2, pad_z = 2;
    BilateralGrid g;
    g.nx = (int)((float)(cols - 1) / sigma_s) + 1 + 2 * pad_xy;
    g.ny = (int)((float)(rows - 1) / sigma_s) + 1 + 2 * pad_xy;
    g.nz = (int)(base_delta / sigma_r) + 1 + 2 * pad_z;
    const size_t cells = (size_t)g.nx * g.ny * g.nz;
    std::vector<long long> sum(cells, 0);
    std::vector<int> cnt(cells, 0);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            float pz = xyz[3 * ((size_t)y * cols + x) + 2];
            if (!std::isfinite(pz)) pz = base_max;
            const float z = pz - base_min;
            const int sx = (int)((float)x / sigma_s + 0.5f) + pad_xy, sy = (int)((float)y / sigma_s + 0.5f) + pad_xy;
            const int sz = (int)(z / sigma_r + 0.5f) + pad_z;
            const size_t c = g.idx(sx, sy, sz);
            sum[c] += (long long)llrint((double)pz * kBilatFixed);
            cnt[c] += 1;
        }
    std::vector<float> a(cells * 2), b(cells * 2, 0.f);
    for (size_t c = 0; c < cells; ++c) {
        a[2 * c] = (float)((double)sum[c] / kBilatFixed);
        a[2 * c + 1] = (float)cnt[c];
    }
    float *data = a.data(), *buffer = b.data();
    const size_t offs[3] = {(size_t)g.ny * g.nz, (size_t)g.nz, 1};
    for (int dim = 0; dim < 3; ++dim)
        for (int it = 0; it < 2; ++it) {
            std::swap(data, buffer);
            const size_t off = offs[dim];
            for (int x = 1; x < g.nx - 1; ++x)
                for (int y = 1; y < g.ny - 1; ++y)
                    for (int z = 1; z < g.nz - 1; ++z) {
                        const size_t c = g.idx(x, y, z);
                        for (int k = 0; k < 2; ++k)
                            data[2 * c + k] = (buffer[2 * (c - off) + k] + buffer[2 * (c + off) + k] + 2.f * buffer[2 * c + k]) / 4.f;
                    }
        }
    auto clampi = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const size_t i = (size_t)y * cols + x;
            float pz = xyz[3 * i + 2];
            if (!std::isfinite(pz)) pz = base_max;
            const float fx = (float)x / sigma_s + (float)pad_xy, fy = (float)y / sigma_s + (float)pad_xy;
            const float fz = (pz - base_min) / sigma_r + (float)pad_z;
            const int x0 = clampi((int)fx, g.nx - 1), x1 = clampi(x0 + 1, g.nx - 1);
            const int y0 = clampi((int)fy, g.ny - 1), y1 = clampi(y0 + 1, g.ny - 1);
            const int z0 = clampi((int)fz, g.nz - 1), z1 = clampi(z0 + 1, g.nz - 1);
            const float xa = fx - (float)x0, ya = fy - (float)y0, za = fz - (float)z0;
            float D[2];
            for (int k = 0; k < 2; ++k)
                D[k] = (1.f - xa) * (1.f - ya) * (1.f - za) * data[2 * g.idx(x0, y0, z0) + k] +
                       xa * (1.f - ya) * (1.f - za) * data[2 * g.idx(x1, y0, z0) + k] +
                       (1.f - xa) * ya * (1.f - za) * data[2 * g.idx(x0, y1, z0) + k] +
                       xa * ya * (1.f - za) * data[2 * g.idx(x1, y1, z0) + k] +
                       (1.f - xa) * (1.f - ya) * za * data[2 * g.idx(x0, y0, z1) + k] +
                       xa * (1.f - ya) * za * data[2 * g.idx(x1, y0, z1) + k] +
                       (1.f - xa) * ya * za * data[2 * g.idx(x0, y1, z1) + k] +
                       xa * ya * za * data[2 * g.idx(x1, y1, z1) + k];
            out[3 * i + 2] = D[0] / D[1];
        }
}

// ------------------------------------------------------------------------------------
// Frame360::stitchSphericalImage / stitchImage (Frame360.h:386-405, 1099-1148): the step right before the alignment
// path.  rgb[s]: sensor_rows x sensor_cols x 3 uint8, depth[s]: uint16 mm; Rt_inv: 8 column-major
// 4x4 (Calib360::Rt_inv, Calib360.h:129); K = {fx, fy, cx, cy} (Calib360.h:74-77).  Outputs: H x W x 3 and H x W with
// W = sensor_rows*8, H = int(W*0.5*60/180), zero where no sensor pixel projects.
// ------------------------------------------------------------------------------------
void oracle_stitch_sphere(const uint8_t* const* rgb, const uint16_t* const* depth, int sensor_rows, int sensor_cols,
                          const float* Rt_inv, const float* K, uint8_t* sphereRGB, uint16_t* sphereDepth) {
    const double PI = 3.14159265359;
    const int W = sensor_rows * 8;
    const int H = (int)(W * 0.5 * 60.0 / 180);
    memset(sphereRGB, 0, (size_t)H * W * 3);
    memset(sphereDepth, 0, (size_t)H * W * sizeof(uint16_t));
    for (int sensor_id = 0; sensor_id < 8; ++sensor_id) {
        const float* M = Rt_inv + 16 * sensor_id;
        const int size_w = sensor_cols, size_h = sensor_rows;
        const float offsetPhi = H / 2 - 0.5;
        const float offsetTheta = -sensor_rows * 15 / 2 + 0.5;
        const float angle_pixel = 2 * PI / W;
        for (int row_phi = 0; row_phi < H; ++row_phi) {
            const float phi_i = (offsetPhi - row_phi) * angle_pixel;
            const float v0 = sinf(phi_i);
            const float cos_phi = cosf(phi_i);
            const int init_col_sphere = (7 - sensor_id) * size_h, end_col_sphere = (8 - sensor_id) * size_h;
            for (int col_theta = init_col_sphere; col_theta < end_col_sphere; ++col_theta) {
                const float theta_i = (col_theta + offsetTheta) * angle_pixel;
                const float v1 = cos_phi * sinf(theta_i);
                const float v2 = cos_phi * cosf(theta_i);
                float p[3];
                for (int i = 0; i < 3; ++i) p[i] = ((M[0 * 4 + i] * v0 + M[1 * 4 + i] * v1) + M[2 * 4 + i] * v2) + M[3 * 4 + i];
                const float u = K[0] * p[0] / p[2] + K[2];
                const float v = K[1] * p[1] / p[2] + K[3];
                if (u >= 0 && u < size_w && v >= 0 && v < size_h) {
                    const int ui = (int)u, vi = (int)v;
                    const size_t o = (size_t)row_phi * W + col_theta, in = (size_t)vi * size_w + ui;
                    for (int k = 0; k < 3; ++k) sphereRGB[3 * o + k] = rgb[sensor_id][3 * in + k];
                    sphereDepth[o] = depth[sensor_id][in] * sqrt(1 + pow((u - K[2]) / K[0], 2) + pow((v - K[3]) / K[1], 2));
                }
            }
        }
    }
}

}  // extern "C"

