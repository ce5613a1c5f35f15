// pbmap_fuzz.cpp -- the host matcher / pose fit of rgbd360_amd/csrc/pbmap_register.h on seeded random and hostile plane lists
// (empty lists, NaN / inf / zero fields, identical planes, more planes than the node budget can explore), meant to be built
// with -fsanitize=address,undefined (tests/test_pbmap_register.py does): the header is host-only C++, so the CPU sanitizers
// see exactly the code that ships inside the library.  Prints a checksum of the results; exit code 0 = no crash and every
// result well formed (status in {0,1,2}, matches injective and in range, finite pose on status 0).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>

#include "../rgbd360_amd/csrc/pbmap_register.h"

static uint64_t s_rng = 88172645463325252ull;
static double urand() {
    s_rng ^= s_rng << 13; s_rng ^= s_rng >> 7; s_rng ^= s_rng << 17;
    return (double)(s_rng >> 11) / 9007199254740992.0;
}
static float hostile(float v) {
    const double u = urand();
    if (u < 0.02) return std::numeric_limits<float>::quiet_NaN();
    if (u < 0.04) return std::numeric_limits<float>::infinity();
    if (u < 0.06) return 0.f;
    if (u < 0.08) return -v;
    return v;
}
static rgbd360_plane random_plane(bool nasty) {
    rgbd360_plane p{};
    double n[3], len = 0;
    for (int k = 0; k < 3; ++k) { n[k] = urand() * 2 - 1; len += n[k] * n[k]; }
    len = std::sqrt(len) + 1e-12;
    for (int k = 0; k < 3; ++k) {
        p.normal[k] = (float)(n[k] / len);
        p.centroid[k] = (float)(urand() * 8 - 4);
        p.ppal_dir[k] = 0.f;
    }
    p.d = -(p.normal[0] * p.centroid[0] + p.normal[1] * p.centroid[1] + p.normal[2] * p.centroid[2]);
    p.curvature = (float)(urand() * 0.002);
    p.area = (float)(0.05 + urand() * 20);
    p.elongation = (float)(1 + urand() * 7);
    p.count = 100;
    if (nasty) {
        for (int k = 0; k < 3; ++k) { p.normal[k] = hostile(p.normal[k]); p.centroid[k] = hostile(p.centroid[k]); }
        p.d = hostile(p.d); p.curvature = hostile(p.curvature); p.area = hostile(p.area); p.elongation = hostile(p.elongation);
    }
    return p;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 400;
    double checksum = 0;
    for (int it = 0; it < rounds; ++it) {
        const bool nasty = it % 3 == 1;
        const int n_ref = it % 17 == 0 ? 0 : (int)(urand() * (it % 50 == 7 ? 60 : 14));
        const int n_trg = it % 19 == 0 ? 0 : (int)(urand() * (it % 50 == 7 ? 60 : 14));
        std::vector<rgbd360_plane> ref, trg;
        for (int i = 0; i < n_ref; ++i) ref.push_back(random_plane(nasty));
        for (int j = 0; j < n_trg; ++j) trg.push_back(it % 5 == 0 && j < n_ref ? ref[j] : random_plane(nasty));   // identical frames too
        if (it % 50 == 7)                    // many mutually consistent planes: the search must stop at its node budget
            for (auto* v : {&ref, &trg})
                for (auto& p : *v) { p.normal[0] = 1; p.normal[1] = p.normal[2] = 0; p.area = 1; p.elongation = 1; p.curvature = 0;
                                     p.centroid[0] = p.centroid[1] = p.centroid[2] = 1; p.d = -1; }
        rgbd360_pbmap_params P;
        pbm::default_params(&P, it & 1);
        if (it % 50 == 7) P.max_nodes = 20000;
        const int mode = it % 4, mmp = it % 7 == 0 ? 5 : 0;
        float pose[16], info[36], area = 0;
        int nm = -1;
        std::vector<int32_t> match(n_ref + 1, -2);
        const int st = pbm::register_planes(ref.data(), n_ref, trg.data(), n_trg, mmp, mode, &P, pose, info, match.data(), &nm, &area);
        if (st < 0 || st > 2 || nm < 0 || nm > n_ref || nm > n_trg) { std::printf("bad status / count at %d: %d %d\n", it, st, nm); return 1; }
        std::vector<char> used(n_trg + 1, 0);
        int seen = 0;
        for (int i = 0; i < n_ref; ++i) {
            if (match[i] == -1) continue;
            if (match[i] < 0 || match[i] >= n_trg || used[match[i]]) { std::printf("bad match at %d\n", it); return 1; }
            used[match[i]] = 1;
            ++seen;
        }
        if (seen != nm) { std::printf("match count mismatch at %d\n", it); return 1; }
        if (st == 0)
            for (int k = 0; k < 16; ++k)
                if (!std::isfinite(pose[k])) { std::printf("non-finite pose at %d\n", it); return 1; }
        checksum += st * 1000 + nm + (st == 0 ? pose[12] : 0);
        // Frame360::mergePlanes on the same (hostile) list: never more planes than it was given, every output well formed
        const pbm::MergeParams M{0.0013f, 0.99f, 0.45f, 0.3f, 0.06f, 0.12f, 6.f};
        const std::vector<rgbd360_plane> merged = pbm::merge_planes(ref.data(), n_ref, M);
        if ((int)merged.size() > n_ref) { std::printf("merge grew the list at %d\n", it); return 1; }
        long long total_in = 0, total_out = 0;
        for (const auto& p : ref) if (pbm::well_formed(p) && !(p.area < M.min_area) && !(p.elongation > M.max_elongation)) total_in += p.count > 0 ? p.count : 1;
        for (const auto& p : merged) {
            if (!std::isfinite(p.d) || !std::isfinite(p.normal[0]) || !std::isfinite(p.centroid[2])) { std::printf("merge produced a non-finite plane at %d\n", it); return 1; }
            total_out += p.count > 0 ? p.count : 1;
        }
        if (total_in != total_out) { std::printf("merge lost inliers at %d: %lld -> %lld\n", it, total_in, total_out); return 1; }
        checksum += (double)merged.size();
    }
    std::printf("ok %d rounds checksum %.6f\n", rounds, checksum);
    return 0;
}
