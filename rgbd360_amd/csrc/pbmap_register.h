// pbmap_register.h -- host side of RegisterRGBD360::RegisterPbMap (RegisterRGBD360.h:110-338 of EduFdez/rgbd360):
// subgraph selection, interpretation-tree plane matching, closed-form pose + information matrix of the matched planes.
// The matcher and the pose fit are mrpt::pbmap (SubgraphMatcher::compareSubgraphs, ConsistencyTest::
// estimatePoseWithCovariance) in the reference -- third-party code that is not in the reference tree; both are restated
// here from the published method (Fernandez-Moral et al., ICRA 2013) with the thresholds of the reference's own
// config_files/configLocaliser_spherical*.ini.  Small data (tens of planes): plain C++ on the host, float64 inside.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/rgbd360_hip.h"

namespace pbm {

// eigen-decomposition of a symmetric 3x3 by cyclic Jacobi (float64): A -> diag(evals), V columns = eigenvectors
inline void jacobi3(const double C[3][3], double evals[3], double V[3][3]) {
    double A[3][3];
    memcpy(A, C, sizeof(A));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[i][j] = i == j;
    // Sweeps until the off-diagonal part is below 1e-26 of the diagonal's size: a rotation by less than that changes no double any more.
    // (Down to 1e-300, i.e. to underflow, it took 9-10 sweeps where 4-5 do: 4.5 us per region, 150 of the 370 us of a frame_planes call.)
    const double scale = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-300 || off <= 1e-26 * scale) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2 * A[p][q]);
                const double t = (theta >= 0 ? 1 : -1) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), sn = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - sn * akq;
                    A[k][q] = sn * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - sn * aqk;
                    A[q][k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - sn * vkq;
                    V[k][q] = sn * vkp + c * vkq;
                }
            }
    }
    for (int k = 0; k < 3; ++k) evals[k] = A[k][k];
}

struct V3 {
    double x, y, z;
};
inline V3 v3(const float* f) { return {f[0], f[1], f[2]}; }
inline double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 sub(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 cross(const V3& a, const V3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(const V3& a) { return sqrt(dot(a, a)); }
inline double clamp1(double c) { return c > 1 ? 1 : (c < -1 ? -1 : c); }
inline double ratio(double a, double b) {      // max / min of two non-negative numbers, +inf when one vanishes
    const double lo = std::min(a, b), hi = std::max(a, b);
    return lo > 0 ? hi / lo : (hi > 0 ? INFINITY : 1.0);
}

// The point that stands for a plane in the distance constraints: the mass centre of its hull polygon -- what
// mrpt::pbmap::Plane::computeMassCenterAndArea leaves in v3center before the plane enters the PbMap (Frame360.h:1028) -- when the
// record carries one, else the inlier centroid (caller-made records).
inline V3 center_of(const rgbd360_plane& p) { return p.hull_points > 0 ? v3(p.center_hull) : v3(p.centroid); }
// the in-plane second moments' rectangle: what a piece's covariance is rebuilt from (area_moment; `area` in records without one)
inline double moment_area(const rgbd360_plane& p) { return p.area_moment > 0 ? p.area_moment : p.area; }

// a record the matcher can use: finite geometry, a unit-length normal, non-negative extent
inline bool well_formed(const rgbd360_plane& p) {
    double nn = 0;
    for (int k = 0; k < 3; ++k) {
        if (!std::isfinite(p.normal[k]) || !std::isfinite(p.centroid[k])) return false;
        nn += (double)p.normal[k] * p.normal[k];
    }
    return std::isfinite(p.d) && std::isfinite(p.curvature) && std::isfinite(p.area) && p.area >= 0 && std::isfinite(p.elongation) &&
           fabs(nn - 1.0) < 1e-3;
}

// setReference / setTarget (RegisterRGBD360.h:110-195): indices of the planes that enter the matching
inline std::vector<int> select_subgraph(const rgbd360_plane* pl_all, int n_all, int max_match_planes, const rgbd360_pbmap_params* P) {
    std::vector<int> kept;                  // the frame's PbMap: Frame360.h:1034,1041 never store small or narrow planes
    for (int i = 0; i < n_all; ++i)
        if (well_formed(pl_all[i]) && !(pl_all[i].area < P->min_area_plane) && !(pl_all[i].elongation > P->max_elongation_plane))
            kept.push_back(i);
    const int n = (int)kept.size();
    const float max_curvature = P->max_curvature_plane;
    std::vector<int> idx;
    if (max_match_planes > 0 && n > max_match_planes) {
        std::vector<float> area(n, 0.f);
        for (int i = 0; i < n; ++i)
            if (pl_all[kept[i]].curvature < max_curvature) area[i] = pl_all[kept[i]].area;
        std::vector<float> sorted = area;
        std::sort(sorted.begin(), sorted.end());
        const float thr = sorted[n - max_match_planes - 1];      // :136: planes strictly above the (n - max - 1)-th area
        for (int i = 0; i < n; ++i)
            if (area[i] > thr) idx.push_back(kept[i]);
    } else {
        for (int i = 0; i < n; ++i)
            if (pl_all[kept[i]].curvature < max_curvature) idx.push_back(kept[i]);
    }
    return idx;
}

struct Matcher {
    const rgbd360_plane* ref;
    const rgbd360_plane* trg;
    std::vector<int> ri, ti;                // subgraph plane indices
    const rgbd360_pbmap_params* P;
    int mode;
    std::vector<signed char> unary;         // [ri.size()][ti.size()]
    std::vector<int> cur, best;             // per ref subgraph plane: matched trg subgraph position or -1
    int cur_n = 0, best_n = -1;
    double best_area = -1;
    std::vector<char> used;
    long long nodes = 0;
    bool out_of_budget = false;

    // Radiometric part of the unary test (configLocaliser_spherical.ini:19-21: color_threshold, intensity_threshold, hue_threshold;
    // mrpt::pbmap::SubgraphMatcher::evalUnaryConstraints, third-party): two planes that both carry colour must agree in every channel
    // of the normalised colour -- invariant to a global brightness change -- in mean intensity (a loose bound), and, when asked for, in
    // their saturated-hue histograms (Bhattacharyya distance).  Planes without colour (color_count = 0) pass.
    bool eval_color(const rgbd360_plane& a, const rgbd360_plane& b) const {
        if (!P->use_color || a.color_count <= 0 || b.color_count <= 0) return true;
        // the DOMINANT colour where both records carry one (what calcMainColor2 leaves in v3colorNrgb / dominantIntensity), else the mean
        const bool mode = a.color_mode_count > 0 && b.color_mode_count > 0;
        const float *ca = mode ? a.color_mode : a.color_nrgb, *cb = mode ? b.color_mode : b.color_nrgb;
        const double ia = mode ? a.intensity_mode : a.intensity, ib = mode ? b.intensity_mode : b.intensity;
        for (int k = 0; k < 3; ++k)
            if (!(fabs((double)ca[k] - cb[k]) < P->color_threshold)) return false;
        if (P->intensity_threshold > 0 && !(fabs(ia - ib) < P->intensity_threshold)) return false;
        if (P->hue_threshold > 0) {
            double bc = 0;
            for (int k = 0; k < 74; ++k) bc += sqrt((double)a.hist_h[k] * (double)b.hist_h[k]);
            if (!(sqrt(std::max(0.0, 1.0 - bc)) < P->hue_threshold)) return false;
        }
        return true;
    }
    bool eval_unary(const rgbd360_plane& a, const rgbd360_plane& b) const {
        if (!(ratio(a.area, b.area) < P->area_threshold)) return false;
        if (!(ratio(a.elongation, b.elongation) < P->elongation_threshold)) return false;
        if (!eval_color(a, b)) return false;
        const V3 na = v3(a.normal), nb = v3(b.normal);
        if (mode == 2 || mode == 3) {           // odometry: small displacement between the two frames
            if (!(dot(na, nb) > cos(P->angle_deg * M_PI / 180))) return false;
            if (!(fabs((double)a.d - b.d) < P->dist_d)) return false;
        }
        if (mode == 1 || mode == 3) {           // planar movement: rotation about the up axis, no change of height
            const double ua = a.normal[P->up_axis], ub = b.normal[P->up_axis];
            if (!(fabs(ua - ub) < P->planar_normal_tol)) return false;
            if (fabs(ua) > 0.98 && !(fabs((double)a.d - b.d) < P->dist_d)) return false;      // floor / ceiling
        }
        return true;
    }
    // pair (a1, a2) of the reference against pair (b1, b2) of the target
    bool eval_binary(const rgbd360_plane& a1, const rgbd360_plane& a2, const rgbd360_plane& b1, const rgbd360_plane& b2) const {
        const V3 na1 = v3(a1.normal), na2 = v3(a2.normal), nb1 = v3(b1.normal), nb2 = v3(b2.normal);
        const double ang_a = acos(clamp1(dot(na1, na2))), ang_b = acos(clamp1(dot(nb1, nb2)));
        if (!(fabs(ang_a - ang_b) < P->angle_threshold_deg * M_PI / 180)) return false;
        const V3 ca = sub(center_of(a2), center_of(a1)), cb = sub(center_of(b2), center_of(b1));
        if (!(ratio(norm(ca), norm(cb)) < P->dist_threshold)) return false;
        if (!(fabs(dot(na1, ca) - dot(nb1, cb)) < P->height_threshold)) return false;         // centre 2 over plane 1
        if (!(fabs(dot(na2, ca) - dot(nb2, cb)) < P->height_threshold)) return false;         // centre 1 under plane 2
        return true;
    }
    // A rigid motion keeps the orientation of every triple of normals: n1 . (n2 x n3) has the same sign in both frames.  A set
    // of matches that violates this is a mirror image of the scene (every angle and distance constraint above is blind to it) and
    // cannot be fitted by any pose.  Not part of the published matcher: it only discards interpretations the pose fit would
    // reject anyway, so that a consistent one can win.  Near-coplanar triples (|triple product| < 0.1) say nothing.
    bool handedness_ok(int k, int j) const {
        const V3 ak = v3(ref[ri[k]].normal), bk = v3(trg[ti[j]].normal);
        for (int m1 = 0; m1 < k; ++m1) {
            if (cur[m1] < 0) continue;
            const V3 a1 = v3(ref[ri[m1]].normal), b1 = v3(trg[ti[cur[m1]]].normal);
            for (int m2 = m1 + 1; m2 < k; ++m2) {
                if (cur[m2] < 0) continue;
                const double ta = dot(a1, cross(v3(ref[ri[m2]].normal), ak));
                const double tb = dot(b1, cross(v3(trg[ti[cur[m2]]].normal), bk));
                if (fabs(ta) > 0.1 && fabs(tb) > 0.1 && (ta > 0) != (tb > 0)) return false;
            }
        }
        return true;
    }
    void explore(int k, double area) {      // area = matched reference area so far, summed in matching order
        if (out_of_budget) return;
        if (P->max_nodes > 0 && ++nodes > P->max_nodes) {
            out_of_budget = true;
            return;
        }
        const int nr = (int)ri.size(), nt = (int)ti.size();
        if (k == nr) {
            if (cur_n > best_n || (cur_n == best_n && area > best_area)) {
                best = cur;
                best_n = cur_n;
                best_area = area;
            }
            return;
        }
        if (cur_n + (nr - k) < best_n) return;          // cannot reach the best count any more
        for (int j = 0; j < nt; ++j) {
            if (used[j] || unary[(size_t)k * nt + j] <= 0) continue;
            bool ok = true;
            for (int m = 0; m < k && ok; ++m)
                if (cur[m] >= 0) ok = eval_binary(ref[ri[m]], ref[ri[k]], trg[ti[cur[m]]], trg[ti[j]]);
            if (!ok || !handedness_ok(k, j)) continue;
            used[j] = 1;
            cur[k] = j;
            ++cur_n;
            explore(k + 1, area + (double)ref[ri[k]].area);
            --cur_n;
            cur[k] = -1;
            used[j] = 0;
        }
        explore(k + 1, area);                            // plane k stays unmatched
    }
    void run() {
        const int nr = (int)ri.size(), nt = (int)ti.size();
        unary.assign((size_t)nr * nt, 0);
        for (int i = 0; i < nr; ++i)
            for (int j = 0; j < nt; ++j) unary[(size_t)i * nt + j] = eval_unary(ref[ri[i]], trg[ti[j]]) ? 1 : -1;
        cur.assign(nr, -1);
        best.assign(nr, -1);
        used.assign(nt, 0);
        explore(0, 0.0);
        if (best_n < 0) best_n = 0;
        if (best_area < 0) best_area = 0;
    }
};

// closed-form pose of matched planes: R = argmax sum w n_ref . (R n_trg), t = argmin sum w (n_ref . t - (d_trg - d_ref))^2.
// pairs: (ref index, trg index).  Returns 0 ok, 2 not observable / inconsistent.
inline int fit_pose(const rgbd360_plane* ref, const rgbd360_plane* trg, const std::vector<std::pair<int, int>>& pairs,
                    const rgbd360_pbmap_params* P, double R[3][3], double t[3], double info[6][6]) {
    double M[3][3] = {{0}}, MtM[3][3] = {{0}};
    for (const auto& pr : pairs) {
        const rgbd360_plane &a = ref[pr.first], &b = trg[pr.second];
        const double w = b.area;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) M[i][j] += w * (double)a.normal[i] * (double)b.normal[j];
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) MtM[i][j] += M[k][i] * M[k][j];
    double ev[3], V[3][3];
    jacobi3(MtM, ev, V);
    int o[3] = {0, 1, 2};
    std::sort(o, o + 3, [&](int a, int b) { return ev[a] > ev[b]; });
    const double s1 = sqrt(std::max(ev[o[0]], 0.0)), s2 = sqrt(std::max(ev[o[1]], 0.0));
    if (!(s1 > 0) || !(s2 > 1e-6 * s1)) return 2;         // all normals parallel: rotation not observable
    V3 v1 = {V[0][o[0]], V[1][o[0]], V[2][o[0]]}, v2 = {V[0][o[1]], V[1][o[1]], V[2][o[1]]};
    auto mul = [&](const V3& v) { return V3{M[0][0] * v.x + M[0][1] * v.y + M[0][2] * v.z, M[1][0] * v.x + M[1][1] * v.y + M[1][2] * v.z,
                                            M[2][0] * v.x + M[2][1] * v.y + M[2][2] * v.z}; };
    V3 u1 = mul(v1), u2 = mul(v2);
    const double n1 = norm(u1);
    u1 = {u1.x / n1, u1.y / n1, u1.z / n1};
    const double p = dot(u1, u2);
    u2 = {u2.x - p * u1.x, u2.y - p * u1.y, u2.z - p * u1.z};
    const double n2 = norm(u2);
    if (!(n2 > 0)) return 2;
    u2 = {u2.x / n2, u2.y / n2, u2.z / n2};
    const V3 u3 = cross(u1, u2), v3c = cross(v1, v2);    // both bases right-handed: det R = +1 (the SVD's reflection fix)
    const double U[3][3] = {{u1.x, u2.x, u3.x}, {u1.y, u2.y, u3.y}, {u1.z, u2.z, u3.z}};
    const double W[3][3] = {{v1.x, v2.x, v3c.x}, {v1.y, v2.y, v3c.y}, {v1.z, v2.z, v3c.z}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            R[i][j] = 0;
            for (int k = 0; k < 3; ++k) R[i][j] += U[i][k] * W[j][k];
        }
    // translation: sum w n n^T t = sum w n (d_trg - d_ref), n = n_ref
    double H[3][3] = {{0}}, g[3] = {0, 0, 0};
    for (const auto& pr : pairs) {
        const rgbd360_plane &a = ref[pr.first], &b = trg[pr.second];
        const double w = b.area, e = (double)b.d - (double)a.d;
        for (int i = 0; i < 3; ++i) {
            g[i] += w * a.normal[i] * e;
            for (int j = 0; j < 3; ++j) H[i][j] += w * (double)a.normal[i] * (double)a.normal[j];
        }
    }
    double hv[3], HV[3][3];
    jacobi3(H, hv, HV);
    const double hmax = std::max(hv[0], std::max(hv[1], hv[2])), hmin = std::min(hv[0], std::min(hv[1], hv[2]));
    if (!(hmin > 0) || !(hmax / hmin < P->max_conditioning)) return 2;   // "Bad conditioning": < 3 independent normals
    for (int i = 0; i < 3; ++i) {
        t[i] = 0;
        for (int k = 0; k < 3; ++k) {
            const double proj = (HV[0][k] * g[0] + HV[1][k] * g[1] + HV[2][k] * g[2]) / hv[k];
            t[i] += HV[i][k] * proj;
        }
    }
    for (int i = 0; i < 3; ++i)
        if (!std::isfinite(t[i]) || !std::isfinite(R[i][0]) || !std::isfinite(R[i][1]) || !std::isfinite(R[i][2])) return 2;
    // consistency of the fit + information matrix blockdiag(sum w n n^T / sigma_d^2, sum w (I - n n^T) / sigma_n^2), n = R n_trg
    memset(info, 0, sizeof(double) * 36);
    const double wd = 1.0 / ((double)P->sigma_dist * P->sigma_dist), wn = 1.0 / ((double)P->sigma_normal * P->sigma_normal);
    for (const auto& pr : pairs) {
        const rgbd360_plane &a = ref[pr.first], &b = trg[pr.second];
        double n[3];
        for (int i = 0; i < 3; ++i) n[i] = R[i][0] * b.normal[0] + R[i][1] * b.normal[1] + R[i][2] * b.normal[2];
        const double c = n[0] * a.normal[0] + n[1] * a.normal[1] + n[2] * a.normal[2];
        if (!(c > P->cos_normal_threshold)) return 2;
        const double w = b.area;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                info[i][j] += w * wd * n[i] * n[j];
                info[3 + i][3 + j] += w * wn * ((i == j) - n[i] * n[j]);
            }
    }
    return 0;
}

inline int register_planes(const rgbd360_plane* ref, int n_ref, const rgbd360_plane* trg, int n_trg, int max_match_planes, int mode,
                           const rgbd360_pbmap_params* P, float pose_out[16], float info_out[36], int32_t* match_out,
                           int* n_matched_out, float* area_matched_out) {
    if (n_ref < 0 || n_trg < 0 || (n_ref > 0 && !ref) || (n_trg > 0 && !trg) || mode < 0 || mode > 3 || !P || P->up_axis < 0 ||
        P->up_axis > 2 || max_match_planes < 0)
        return -1;
    if (pose_out)
        for (int i = 0; i < 16; ++i) pose_out[i] = (i % 5 == 0) ? 1.f : 0.f;
    if (info_out) memset(info_out, 0, 36 * sizeof(float));
    if (match_out)
        for (int i = 0; i < n_ref; ++i) match_out[i] = -1;
    Matcher m;
    m.ref = ref;
    m.trg = trg;
    m.P = P;
    m.mode = mode;
    m.ri = select_subgraph(ref, n_ref, max_match_planes, P);
    m.ti = select_subgraph(trg, n_trg, max_match_planes, P);
    m.run();
    std::vector<std::pair<int, int>> pairs;
    for (size_t k = 0; k < m.ri.size(); ++k)
        if (m.best[k] >= 0) pairs.emplace_back(m.ri[k], m.ti[m.best[k]]);
    if (match_out)
        for (const auto& pr : pairs) match_out[pr.first] = pr.second;
    if (n_matched_out) *n_matched_out = (int)pairs.size();
    if (area_matched_out) *area_matched_out = (float)m.best_area;
    if ((int)pairs.size() < P->min_planes_recognition || pairs.size() < 3) return 1;     // RegisterRGBD360.h:312
    double R[3][3], t[3], info[6][6];
    const int st = fit_pose(ref, trg, pairs, P, R, t, info);
    if (st != 0) return st;
    if (pose_out) {
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) pose_out[j * 4 + i] = (float)R[i][j];
            pose_out[12 + i] = (float)t[i];
        }
    }
    if (info_out)
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) info_out[j * 6 + i] = (float)info[i][j];
    return 0;
}

// ---- Frame360::mergePlanes (Frame360.h:655-733): the co-planar pieces several sensors see of one surface become one plane ----
// The reference's test for "same surface" is explicit -- normals' dot product > 0.99, |d_j - d_k| < 0.45 m, and two contour
// vertices closer than 0.3 m whose difference is within 0.06 m of plane j (or two contour edges closer than 0.3 m) -- but runs on
// the convex hulls mrpt::pbmap keeps; here the "contour" of a plane is the rectangle with its in-plane moments (centre c,
// half-extents sqrt(3 l2) along ppal_dir and sqrt(3 l1) across): its four corners, four edge midpoints and centre stand in for the
// hull vertices, and a point of one rectangle lying inside the other (within 0.06 m of its plane) stands in for crossing edges.
// mrpt's mergePlane2 pools the inliers and refits; the pooled fit is exact here: the covariance of every piece is rebuilt from its
// record (l0 from the curvature, l1 / l2 from area and elongation, axes normal / ppal_dir), the pieces are combined by count.
struct PlaneMoments {
    double n, c[3], C[3][3];
};
inline PlaneMoments moments_of(const rgbd360_plane& p) {
    PlaneMoments m;
    m.n = p.count > 0 ? p.count : 1;
    V3 nn = v3(p.normal), pp = v3(p.ppal_dir);
    if (!(norm(pp) > 0.5)) {                              // no principal direction in the record: any unit vector across the normal
        const V3 e = fabs(nn.x) < 0.9 ? V3{1, 0, 0} : V3{0, 1, 0};
        pp = cross(nn, e);
    }
    const double pn = dot(pp, nn);
    pp = {pp.x - pn * nn.x, pp.y - pn * nn.y, pp.z - pn * nn.z};
    const double pl = norm(pp);
    pp = {pp.x / pl, pp.y / pl, pp.z / pl};
    const V3 qq = cross(nn, pp);
    const double el = p.elongation > 0 && std::isfinite(p.elongation) ? p.elongation : 1.0;
    const double am = moment_area(p);
    const double l1 = am / (12.0 * el), l2 = am * el / 12.0;
    const double cv = p.curvature < 0.5 ? p.curvature : 0.5;
    const double l0 = cv * (l1 + l2) / (1.0 - cv);
    const double a[3][3] = {{nn.x, nn.y, nn.z}, {qq.x, qq.y, qq.z}, {pp.x, pp.y, pp.z}};
    const double l[3] = {l0, l1, l2};
    for (int i = 0; i < 3; ++i) {
        m.c[i] = p.centroid[i];
        for (int j = 0; j < 3; ++j) {
            m.C[i][j] = 0;
            for (int k = 0; k < 3; ++k) m.C[i][j] += l[k] * a[k][i] * a[k][j];
        }
    }
    return m;
}
inline rgbd360_plane plane_of(const PlaneMoments& m, int root) {
    rgbd360_plane P{};
    double ev[3], V[3][3];
    jacobi3(m.C, ev, V);
    int o[3] = {0, 1, 2};
    std::stable_sort(o, o + 3, [&](int a, int b) { return ev[a] < ev[b]; });
    double nn[3] = {V[0][o[0]], V[1][o[0]], V[2][o[0]]};
    double d = -(nn[0] * m.c[0] + nn[1] * m.c[1] + nn[2] * m.c[2]);
    if (d < 0) {                                           // towards the origin, like the plane fit
        for (int i = 0; i < 3; ++i) nn[i] = -nn[i];
        d = -d;
    }
    const double l0 = std::max(ev[o[0]], 0.0), l1 = std::max(ev[o[1]], 0.0), l2 = std::max(ev[o[2]], 0.0);
    for (int i = 0; i < 3; ++i) {
        P.centroid[i] = (float)m.c[i];
        P.normal[i] = (float)nn[i];
        P.ppal_dir[i] = (float)V[i][o[2]];
    }
    P.d = (float)d;
    P.curvature = (float)(l0 + l1 + l2 > 0 ? l0 / (l0 + l1 + l2) : 0.0);
    P.count = (int)m.n;
    P.root = root;
    P.area = P.area_moment = (float)(12.0 * sqrt(l1 * l2));
    P.elongation = (float)(l1 > 0 ? sqrt(l2 / l1) : INFINITY);
    for (int i = 0; i < 3; ++i) P.center_hull[i] = P.centroid[i];
    P.hull_points = 0;
    P.hull_n = 0;
    return P;
}
struct MergeParams {
    float max_curvature, cos_normal, dist_d, proximity, normal_offset;
    float min_area, max_elongation;        // Frame360.h:1034,1041: smaller / narrower regions are never stored, so they never reach the merge
};
inline void contour_points(const rgbd360_plane& p, V3 pts[9], V3& pp, V3& qq, double& a, double& b) {
    const PlaneMoments m = moments_of(p);                  // (re-derives the axes the same way the merge does)
    const V3 nn = v3(p.normal);
    pp = v3(p.ppal_dir);
    if (!(norm(pp) > 0.5)) pp = cross(nn, fabs(nn.x) < 0.9 ? V3{1, 0, 0} : V3{0, 1, 0});
    const double pn = dot(pp, nn);
    pp = {pp.x - pn * nn.x, pp.y - pn * nn.y, pp.z - pn * nn.z};
    const double pl = norm(pp);
    pp = {pp.x / pl, pp.y / pl, pp.z / pl};
    qq = cross(nn, pp);
    const double el = p.elongation > 0 && std::isfinite(p.elongation) ? p.elongation : 1.0;
    a = sqrt(3.0 * p.area * el / 12.0);
    b = sqrt(3.0 * p.area / (12.0 * el));
    const V3 c = {m.c[0], m.c[1], m.c[2]};
    int k = 0;
    for (int su = -1; su <= 1; ++su)
        for (int sv = -1; sv <= 1; ++sv)
            pts[k++] = {c.x + su * a * pp.x + sv * b * qq.x, c.y + su * a * pp.y + sv * b * qq.y, c.z + su * a * pp.z + sv * b * qq.z};
}
// Squared distance between the 3-D segments [p0, p1] and [q0, q1] (mrpt::pbmap::dist3D_Segment_to_Segment2, Frame360.h:697: the classic
// clamped closest-point solution; MRPT is third-party, restated).
inline double seg_seg_dist2(const V3& p0, const V3& p1, const V3& q0, const V3& q1) {
    const V3 u = sub(p1, p0), v = sub(q1, q0), w = sub(p0, q0);
    const double a = dot(u, u), b = dot(u, v), c = dot(v, v), d = dot(u, w), e = dot(v, w), D = a * c - b * b;
    double sN, sD = D, tN, tD = D;
    if (D < 1e-12 * std::max(a * c, 1e-300)) {             // almost parallel: use p0 on the first segment
        sN = 0.0; sD = 1.0; tN = e; tD = c;
    } else {
        sN = b * e - c * d; tN = a * e - b * d;
        if (sN < 0.0) { sN = 0.0; tN = e; tD = c; }
        else if (sN > sD) { sN = sD; tN = e + b; tD = c; }
    }
    if (tN < 0.0) {
        tN = 0.0;
        if (-d < 0.0) sN = 0.0;
        else if (-d > a) sN = sD;
        else { sN = -d; sD = a; }
    } else if (tN > tD) {
        tN = tD;
        if (-d + b < 0.0) sN = 0.0;
        else if (-d + b > a) sN = sD;
        else { sN = -d + b; sD = a; }
    }
    const double sc = fabs(sN) < 1e-300 || !(sD > 0) ? 0.0 : sN / sD, tc = fabs(tN) < 1e-300 || !(tD > 0) ? 0.0 : tN / tD;
    const V3 dP = {w.x + sc * u.x - tc * v.x, w.y + sc * u.y - tc * v.y, w.z + sc * u.z - tc * v.z};
    return dot(dP, dP);
}
// in-plane frame of a record (unit axes e1, e2 with e1 x e2 = n) and a polygon's vertices in it
inline void plane_axes(const rgbd360_plane& p, V3& e1, V3& e2) {
    const V3 nn = v3(p.normal);
    e1 = v3(p.ppal_dir);
    if (!(norm(e1) > 0.5)) e1 = cross(nn, fabs(nn.x) < 0.9 ? V3{1, 0, 0} : V3{0, 1, 0});
    const double pn = dot(e1, nn);
    e1 = {e1.x - pn * nn.x, e1.y - pn * nn.y, e1.z - pn * nn.z};
    const double l = norm(e1);
    e1 = {e1.x / l, e1.y / l, e1.z / l};
    e2 = cross(nn, e1);
}
inline bool point_in_hull(const rgbd360_plane& p, const V3& q, double normal_offset) {      // q within normal_offset of p's plane and inside its polygon
    const int n = std::min(p.hull_n, (int)RGBD360_HULL_MAX);
    if (n < 3) return false;
    const V3 nn = v3(p.normal);
    if (!(fabs(dot(nn, sub(q, v3(p.hull[0])))) < normal_offset)) return false;
    double sgn = 0;                                        // the same side of every edge (either sense)
    for (int i = 0; i < n; ++i) {
        const V3 a = v3(p.hull[i]), b = v3(p.hull[(i + 1) % n]);
        const double s = dot(nn, cross(sub(b, a), sub(q, a)));
        if (s == 0) continue;
        if (sgn == 0) sgn = s;
        else if ((s > 0) != (sgn > 0)) return false;
    }
    return true;
}
inline bool same_surface_hulls(const rgbd360_plane& pj, const rgbd360_plane& pk, const MergeParams& M) {
    const V3 nj = v3(pj.normal);
    const int nj_v = std::min(pj.hull_n, (int)RGBD360_HULL_MAX), nk_v = std::min(pk.hull_n, (int)RGBD360_HULL_MAX);
    for (int i = 0; i < nj_v; ++i)                         // Frame360.h:680-691: vertex against vertex
        for (int ii = 0; ii < nk_v; ++ii) {
            const V3 df = sub(v3(pj.hull[i]), v3(pk.hull[ii]));
            if (norm(df) < M.proximity && fabs(dot(nj, df)) < M.normal_offset) return true;
        }
    const double prox2 = (double)M.proximity * M.proximity;
    for (int i = 0; i < nj_v; ++i)                         // :694-711: edge against edge (the polygons are closed)
        for (int ii = 0; ii < nk_v; ++ii) {
            const V3 a0 = v3(pj.hull[i]), a1 = v3(pj.hull[(i + 1) % nj_v]), b0 = v3(pk.hull[ii]), b1 = v3(pk.hull[(ii + 1) % nk_v]);
            if (seg_seg_dist2(a0, a1, b0, b1) < prox2 && fabs(dot(nj, sub(a1, b1))) < M.normal_offset) return true;
        }
    // (round 5 also accepted a vertex of one polygon INSIDE the other; the reference has no such test -- Frame360.h:680-711, 788-815 are
    // vertex-vertex and edge-edge proximity only -- and a panel lying inside a wall's hull, farther than `proximity` from its outline,
    // stays a plane of its own there: removed in round 6.)
    return false;
}
inline bool same_surface(const rgbd360_plane& pj, const rgbd360_plane& pk, const MergeParams& M) {
    const V3 nj = v3(pj.normal), nk = v3(pk.normal);
    if (!(dot(nj, nk) > M.cos_normal)) return false;                              // Frame360.h:671
    if (!(fabs((double)pj.d - pk.d) < M.dist_d)) return false;                    // :672
    if (pj.hull_n >= 3 && pk.hull_n >= 3) return same_surface_hulls(pj, pk, M);   // :680-711 on the polygons the records carry
    V3 Pj[9], Pk[9], ppj, qqj, ppk, qqk;
    double aj, bj, ak, bk;
    contour_points(pj, Pj, ppj, qqj, aj, bj);
    contour_points(pk, Pk, ppk, qqk, ak, bk);
    for (int i = 0; i < 9; ++i)
        for (int ii = 0; ii < 9; ++ii) {                                          // :680-691 vertex against vertex
            const V3 df = sub(Pj[i], Pk[ii]);
            if (norm(df) < M.proximity && fabs(dot(nj, df)) < M.normal_offset) return true;
        }
    auto inside = [&](const V3& q, const rgbd360_plane& p, const V3& pp, const V3& qq, double a, double b) {
        const V3 df = sub(q, v3(p.centroid));
        return fabs(dot(df, pp)) <= a && fabs(dot(df, qq)) <= b && fabs(dot(v3(p.normal), df)) < M.normal_offset;
    };
    for (int i = 0; i < 9; ++i)                                                   // :694-711 stand-in: overlapping outlines
        if (inside(Pk[i], pj, ppj, qqj, aj, bj) || inside(Pj[i], pk, ppk, qqk, ak, bk)) return true;
    return false;
}
// The polygon of a merged plane: the convex hull, on the pooled plane, of the two pieces' polygon vertices (what mergePlane2 does with
// the two contours); sets hull / hull_n / area / center_hull of dst.  false: degenerate (fewer than three distinct projected points).
inline bool rehull(rgbd360_plane& dst, const rgbd360_plane& a, const rgbd360_plane& b, const int hull_points_sum) {
    V3 e1, e2;
    plane_axes(dst, e1, e2);
    const V3 c = v3(dst.centroid);
    struct P2 { double x, y; };
    std::vector<P2> pts;
    for (const rgbd360_plane* p : {&a, &b})
        for (int i = 0; i < p->hull_n && i < (int)RGBD360_HULL_MAX; ++i) {
            const V3 d = sub(v3(p->hull[i]), c);
            pts.push_back({dot(d, e1), dot(d, e2)});
        }
    std::sort(pts.begin(), pts.end(), [](const P2& p, const P2& q) { return p.x < q.x || (p.x == q.x && p.y < q.y); });
    pts.erase(std::unique(pts.begin(), pts.end(), [](const P2& p, const P2& q) { return p.x == q.x && p.y == q.y; }), pts.end());
    const int n = (int)pts.size();
    if (n < 3) return false;
    auto cr = [](const P2& o, const P2& p, const P2& q) { return (p.x - o.x) * (q.y - o.y) - (p.y - o.y) * (q.x - o.x); };
    std::vector<P2> H(2 * n + 2);
    int m = 0;
    for (int i = 0; i < n; ++i) {
        while (m >= 2 && cr(H[m - 2], H[m - 1], pts[i]) <= 0) --m;
        H[m++] = pts[i];
    }
    for (int i = n - 2, lo = m + 1; i >= 0; --i) {
        while (m >= lo && cr(H[m - 2], H[m - 1], pts[i]) <= 0) --m;
        H[m++] = pts[i];
    }
    --m;
    if (m < 3) return false;
    double a2 = 0, cu = 0, cv = 0;
    for (int i = 0; i < m; ++i) {
        const P2 &p = H[i], &q = H[(i + 1) % m];
        const double w = p.x * q.y - p.y * q.x;
        a2 += w; cu += (p.x + q.x) * w; cv += (p.y + q.y) * w;
    }
    if (!(fabs(a2) > 0)) return false;
    cu /= 3 * a2; cv /= 3 * a2;
    // at most 2 x RGBD360_HULL_MAX inputs: thin to the extremes in RGBD360_HULL_MAX directions when the hull is larger (apply_hull's rule)
    std::vector<int> keep;
    if (m <= (int)RGBD360_HULL_MAX) {
        for (int i = 0; i < m; ++i) keep.push_back(i);
    } else {
        for (int k = 0; k < (int)RGBD360_HULL_MAX; ++k) {
            const double th = 2.0 * 3.14159265358979323846 * k / RGBD360_HULL_MAX, cx = cos(th), sy = sin(th);
            int best = 0;
            double bd = -1e300;
            for (int i = 0; i < m; ++i) {
                const double dd = (H[i].x - cu) * cx + (H[i].y - cv) * sy;
                if (dd > bd) { bd = dd; best = i; }
            }
            if ((!keep.empty() && (keep.back() == best || keep.front() == best))) continue;
            keep.push_back(best);
        }
    }
    dst.hull_n = (int)keep.size();
    for (int i = 0; i < dst.hull_n; ++i) {                 // (e1, e2, n) is right-handed: counter-clockwise in (u, v) = counter-clockwise seen from the normal's side
        const P2& h = H[keep[i]];
        dst.hull[i][0] = (float)(c.x + h.x * e1.x + h.y * e2.x);
        dst.hull[i][1] = (float)(c.y + h.x * e1.y + h.y * e2.y);
        dst.hull[i][2] = (float)(c.z + h.x * e1.z + h.y * e2.z);
    }
    dst.area = (float)(fabs(a2) / 2);
    dst.center_hull[0] = (float)(c.x + cu * e1.x + cv * e2.x);
    dst.center_hull[1] = (float)(c.y + cu * e1.y + cv * e2.y);
    dst.center_hull[2] = (float)(c.z + cu * e1.z + cv * e2.z);
    dst.hull_points = hull_points_sum;
    return true;
}
// colour of a merged plane = the pooled statistics of its pieces (mergePlane2 pools the inliers and calls calcMainColor again):
// means and histogram weighted by the pixel counts they were taken over, the deviation from the pooled second moment
inline void pool_colour(rgbd360_plane& dst, const rgbd360_plane& a, const rgbd360_plane& b) {
    const double na = std::max(a.color_count, 0), nb = std::max(b.color_count, 0);
    dst.color_count = (int)(na + nb);
    if (na + nb > 0) {
        for (int k = 0; k < 3; ++k) {
            const double m = (na * a.color_nrgb[k] + nb * b.color_nrgb[k]) / (na + nb);
            const double var = (na * ((double)a.color_dev[k] * a.color_dev[k] + (a.color_nrgb[k] - m) * (a.color_nrgb[k] - m)) +
                                nb * ((double)b.color_dev[k] * b.color_dev[k] + (b.color_nrgb[k] - m) * (b.color_nrgb[k] - m))) / (na + nb);
            dst.color_nrgb[k] = (float)m;
            dst.color_dev[k] = (float)sqrt(std::max(var, 0.0));
        }
        dst.intensity = (float)((na * a.intensity + nb * b.intensity) / (na + nb));
    }
    // the dominant colour of the pooled inliers: mergePlane2 runs calcMainColor2 again on them; the records no longer hold the samples, so
    // the mode of the piece with more samples on its mode stands for it (two pieces of one surface share their dominant colour)
    {
        const double ka = a.color_mode_count > 0 ? (double)a.color_mode_count * a.color_concentration * std::max(a.count, 1) / std::max(a.color_mode_count, 1) : 0;
        const double kb = b.color_mode_count > 0 ? (double)b.color_mode_count * b.color_concentration * std::max(b.count, 1) / std::max(b.color_mode_count, 1) : 0;
        const rgbd360_plane* w = ka >= kb ? &a : &b;
        if (ka > 0 || kb > 0) {
            dst.color_mode_count = a.color_mode_count + b.color_mode_count;
            for (int k = 0; k < 3; ++k) dst.color_mode[k] = w->color_mode[k];
            dst.intensity_mode = w->intensity_mode;
            dst.color_concentration = (float)(std::max(ka, kb) / std::max((double)std::max(a.count, 0) + std::max(b.count, 0), 1.0));
        }
    }
    // (normalised histograms of the pixels that carry colour: pooled with the weights of the means above)
    if (na + nb > 0)
        for (int k = 0; k < 74; ++k) dst.hist_h[k] = (float)((na * a.hist_h[k] + nb * b.hist_h[k]) / (na + nb));
}
// mrpt::pbmap::Plane::mergePlane2 (third-party; Frame360.h:717, 815): the exact pooled fit of the two pieces' moments, the two contours
// pooled and hulled again on the merged plane (area and mass centre are that polygon's), the colour descriptors pooled
inline rgbd360_plane pool_planes(const rgbd360_plane& pj, const rgbd360_plane& pk) {
    const PlaneMoments a = moments_of(pj), b = moments_of(pk);
    PlaneMoments m;
    m.n = a.n + b.n;
    for (int i = 0; i < 3; ++i) m.c[i] = (a.n * a.c[i] + b.n * b.c[i]) / m.n;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            m.C[r][c] = (a.n * (a.C[r][c] + (a.c[r] - m.c[r]) * (a.c[c] - m.c[c])) + b.n * (b.C[r][c] + (b.c[r] - m.c[r]) * (b.c[c] - m.c[c]))) / m.n;
    const bool hulls = pj.hull_points > 0 && pk.hull_points > 0;
    const double aj = pj.area, ak = pk.area;
    const int hp = pj.hull_points + pk.hull_points;
    const V3 cj = center_of(pj), ck = center_of(pk);
    rgbd360_plane out = plane_of(m, std::min(pj.root, pk.root));
    if (pj.hull_n >= 3 && pk.hull_n >= 3 && rehull(out, pj, pk, hp)) {
        // (the polygon of the merged plane)
    } else if (hulls && aj + ak > 0) {
        // records without polygons: the merged surface's area is the pieces' sum, its centre their area-weighted mean
        out.area = (float)(aj + ak);
        out.center_hull[0] = (float)((aj * cj.x + ak * ck.x) / (aj + ak));
        out.center_hull[1] = (float)((aj * cj.y + ak * ck.y) / (aj + ak));
        out.center_hull[2] = (float)((aj * cj.z + ak * ck.z) / (aj + ak));
        out.hull_points = hp;
    }
    pool_colour(out, pj, pk);
    return out;
}
inline std::vector<rgbd360_plane> merge_planes(const rgbd360_plane* in, int n, const MergeParams& M) {
    std::vector<rgbd360_plane> v;
    for (int i = 0; i < n; ++i)
        if (well_formed(in[i]) && !(in[i].area < M.min_area) && !(in[i].elongation > M.max_elongation)) v.push_back(in[i]);
    for (size_t j = 0; j < v.size(); ++j) {
        bool merged = true;
        // :727-731 re-evaluate plane j after every merge (`j--`): the outer curvature test (:663) runs again on the pooled fit, so a
        // plane whose curvature has risen above the limit stops absorbing neighbours
        while (merged && v[j].curvature < M.max_curvature) {
            merged = false;
            for (size_t k = j + 1; k < v.size(); ++k) {
                if (!(v[k].curvature < M.max_curvature) || !same_surface(v[j], v[k], M)) continue;
                v[j] = pool_planes(v[j], v[k]);
                v.erase(v.begin() + (long)k);
                merged = true;
                break;
            }
        }
    }
    return v;
}

// Frame360::getPlanesSensor's tail (Frame360.h:1034-1068), the three steps between a sensor's regions and local_planes_[sensor]:
// regions smaller than min_area (:1034) or narrower than max_elongation (:1041) are never stored; a region flatter than max_curvature
// is pooled (mergePlane2) into the FIRST stored plane, also flatter, that `isSamePlane(plane, 0.99, 0.05, 0.2)` (:1056-1068), else
// appended.  mrpt::pbmap::Plane::isSamePlane / isPlaneNearby are third-party (MRPT 1.x pbmap/Plane.cpp, restated, unpinned): normals
// closer than cos_normal; the other plane's centre within dist_normal of this plane along its normal; the two outlines nearer than
// proximity -- centre to centre, a vertex of one to the other's centre, vertex to vertex, edge to edge (dist3D_Segment_to_Segment2).
struct SensorPoolParams {
    float max_curvature = 0.0013f, min_area = 0.12f, max_elongation = 6.f;      // Miscellaneous.h:54,57,60
    float cos_normal = 0.99f, dist_normal = 0.05f, proximity = 0.2f;            // Frame360.h:1058
};
inline int outline_points(const rgbd360_plane& p, V3 pts[RGBD360_HULL_MAX]) {      // the hull polygon, else the moment rectangle's corners
    const int n = std::min(p.hull_n, (int)RGBD360_HULL_MAX);
    if (n >= 3) {
        for (int i = 0; i < n; ++i) pts[i] = v3(p.hull[i]);
        return n;
    }
    V3 c9[9], pp, qq;
    double a, b;
    contour_points(p, c9, pp, qq, a, b);
    const int corner[4] = {0, 2, 8, 6};                                            // (-,-) (-,+) (+,+) (+,-): a closed outline
    for (int i = 0; i < 4; ++i) pts[i] = c9[corner[i]];
    return 4;
}
inline bool is_plane_nearby(const rgbd360_plane& a, const rgbd360_plane& b, double prox) {
    const double p2 = prox * prox;
    const V3 ca = center_of(a), cb = center_of(b);
    const V3 dc = sub(ca, cb);
    if (dot(dc, dc) < p2) return true;
    V3 A[RGBD360_HULL_MAX], B[RGBD360_HULL_MAX];
    const int na = outline_points(a, A), nb = outline_points(b, B);
    for (int i = 0; i < na; ++i) { const V3 d = sub(A[i], cb); if (dot(d, d) < p2) return true; }
    for (int j = 0; j < nb; ++j) { const V3 d = sub(ca, B[j]); if (dot(d, d) < p2) return true; }
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) { const V3 d = sub(A[i], B[j]); if (dot(d, d) < p2) return true; }
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j)
            if (seg_seg_dist2(A[i], A[(i + 1) % na], B[j], B[(j + 1) % nb]) < p2) return true;
    return false;
}
inline bool is_same_plane(const rgbd360_plane& a, const rgbd360_plane& b, const SensorPoolParams& P) {
    const V3 na = v3(a.normal);
    if (dot(na, v3(b.normal)) < P.cos_normal) return false;
    if (fabs(dot(na, sub(center_of(b), center_of(a)))) > P.dist_normal) return false;
    return is_plane_nearby(a, b, P.proximity);
}
inline std::vector<rgbd360_plane> pool_sensor_planes(const rgbd360_plane* in, int n, const SensorPoolParams& P) {
    std::vector<rgbd360_plane> v;
    for (int i = 0; i < n; ++i) {
        const rgbd360_plane& pl = in[i];
        if (!well_formed(pl) || pl.area < P.min_area || pl.elongation > P.max_elongation) continue;      // :1034, :1041
        bool same = false;
        if (pl.curvature < P.max_curvature)
            for (size_t j = 0; j < v.size() && !same; ++j)
                if (v[j].curvature < P.max_curvature && is_same_plane(v[j], pl, P)) {
                    v[j] = pool_planes(v[j], pl);
                    same = true;
                }
        if (!same) v.push_back(pl);
    }
    return v;
}

// Frame360::groupPlanes (Frame360.h:741-833): the planes of the eight sensors, in sensor order, gathered into the frame's plane list --
// a plane of sensor s is pooled into a plane that came from (or absorbed a piece of) sensor s - 1 when both are large and flat enough
// (:764, :770) and lie on one surface (|d_j - d_k| < dist_d :775, normals :776, hull polygons within max_dist_hull of each other --
// vertex against vertex :788-798, edge against edge :800-815 -- with the offset along the absorbing plane's normal below
// max_dist_parallel_hull), else appended; sensor 7's candidates also include sensor 0's planes (:827-828: the ring closes).
// Planes are NOT filtered here (small or narrow ones stay in the list: mergePlanes / the registration's subgraph selection drop them).
struct GroupParams {
    float max_curvature = 0.0013f;      // max_curvature_plane, Miscellaneous.h:54
    float min_area = 0.5f;              // :764, :770
    float cos_normal = 0.99f, dist_d = 0.45f;
    float max_dist_hull = 0.5f, max_dist_parallel_hull = 0.09f;      // :746-747
};
inline std::vector<rgbd360_plane> group_planes(const rgbd360_plane* in, const int* n_per_sensor, int n_sensors, const GroupParams& G) {
    MergeParams M{};
    M.cos_normal = G.cos_normal; M.dist_d = G.dist_d; M.proximity = G.max_dist_hull; M.normal_offset = G.max_dist_parallel_hull;
    std::vector<rgbd360_plane> v;
    std::vector<int> prev, first;
    int at = 0;
    for (int s = 0; s < n_sensors; ++s) {
        std::vector<int> next_prev;
        for (int k = 0; k < n_per_sensor[s]; ++k) {
            const rgbd360_plane& pk = in[at + k];
            if (!well_formed(pk)) continue;
            int hit = -1;
            if (s > 0 && (pk.area > G.min_area || pk.curvature < G.max_curvature))        // :764 (an OR, as written)
                for (int j : prev) {
                    if (v[(size_t)j].area < G.min_area || v[(size_t)j].curvature > G.max_curvature) continue;      // :770
                    if (same_surface(v[(size_t)j], pk, M)) { hit = j; break; }
                }
            if (hit >= 0) {
                v[(size_t)hit] = pool_planes(v[(size_t)hit], pk);
                next_prev.push_back(hit);
            } else {
                next_prev.push_back((int)v.size());
                v.push_back(pk);
            }
        }
        std::sort(next_prev.begin(), next_prev.end());                                     // (std::set<unsigned>: ascending, unique)
        next_prev.erase(std::unique(next_prev.begin(), next_prev.end()), next_prev.end());
        if (s == 0) first = next_prev;
        prev = next_prev;
        if (s == n_sensors - 2) {                                                          // :827-828 (sensor_id == 6 of 8)
            prev.insert(prev.end(), first.begin(), first.end());
            std::sort(prev.begin(), prev.end());
            prev.erase(std::unique(prev.begin(), prev.end()), prev.end());
        }
        at += n_per_sensor[s];
    }
    return v;
}

inline void default_params(rgbd360_pbmap_params* p, int odometry) {
    // config_files/configLocaliser_spherical.ini (0) / configLocaliser_sphericalOdometry.ini (1)
    p->dist_d = odometry ? 0.5f : 0.4f;
    p->angle_deg = odometry ? 50.f : 40.f;
    p->elongation_threshold = odometry ? 2.5f : 3.8f;
    p->area_threshold = odometry ? 3.0f : 4.0f;
    p->dist_threshold = odometry ? 3.0f : 4.0f;
    p->angle_threshold_deg = odometry ? 10.f : 9.f;
    p->height_threshold = 0.33f;
    p->cos_normal_threshold = odometry ? 0.985f : 0.99f;
    p->min_planes_recognition = 3;
    p->max_curvature_plane = 0.0013f;     // Miscellaneous.h:54
    p->min_area_plane = 0.12f;            // Miscellaneous.h:57
    p->max_elongation_plane = 6.f;        // Miscellaneous.h:60
    p->up_axis = 0;
    p->planar_normal_tol = 0.08f;
    p->max_conditioning = 100.f;
    p->sigma_dist = 0.02f;                // the segmentation's distance threshold (Frame360.h:960-965)
    p->sigma_normal = 0.0398f;            // its angular threshold
    p->max_nodes = 2000000;
    p->use_color = 1;
    p->color_threshold = 0.07f;
    p->intensity_threshold = odometry ? 100.f : 150.f;
    p->hue_threshold = 0.f;               // ini: 0.35 / 0.45, but the reference's own test of it is commented out (Frame360.h:673): opt-in
}

}  // namespace pbm
