// pbmap_register_demo.cpp -- the reference's keyframe-link sequence (KFsphere_SLAM.cpp:182, 314-317: RegisterPbMap ->
// getPose / calcEntropy / getMatchedPlanes / getAreaMatched) through the C++ adapter, on plane lists read from a text file
// (one plane per line: cx cy cz nx ny nz d curvature area elongation; a line "--" separates the two frames).
// Prints: status, matches "i:j", the 4x4 pose row by row, entropy, matched area.  Host only (no GPU work).
//   usage: pbmap_register_demo <planes.txt> <max_match_planes> <regist_mode 0..3> <odometry_config 0|1>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rgbd360/RegisterRGBD360.hpp"

int main(int argc, char** argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s planes.txt max_match_planes regist_mode odometry_config\n", argv[0]);
        return 2;
    }
    FILE* f = std::fopen(argv[1], "r");
    if (!f) return 3;
    std::vector<rgbd360_plane> frames[2];
    int which = 0;
    char line[512];
    while (std::fgets(line, sizeof(line), f)) {
        if (std::strncmp(line, "--", 2) == 0) {
            which = 1;
            continue;
        }
        rgbd360_plane p{};
        if (std::sscanf(line, "%f %f %f %f %f %f %f %f %f %f", &p.centroid[0], &p.centroid[1], &p.centroid[2], &p.normal[0], &p.normal[1],
                        &p.normal[2], &p.d, &p.curvature, &p.area, &p.elongation) == 10)
            frames[which].push_back(p);
    }
    std::fclose(f);
    rgbd360::RegisterRGBD360 registerer(std::atoi(argv[4]) != 0);
    rgbd360::PlaneList ref{frames[0].data(), (int)frames[0].size()}, trg{frames[1].data(), (int)frames[1].size()};
    const bool good = registerer.RegisterPbMap(&ref, &trg, (size_t)std::atoi(argv[2]),
                                               (rgbd360::RegisterRGBD360::registrationType)std::atoi(argv[3]));
    std::printf("status %d good %d\n", registerer.status(), good ? 1 : 0);
    std::printf("matches");
    for (const auto& m : registerer.getMatchedPlanes()) std::printf(" %u:%u", m.first, m.second);
    std::printf("\n");
    const rgbd360::Mat4f T = registerer.getPose();
    for (int r = 0; r < 4; ++r) std::printf("%.9g %.9g %.9g %.9g\n", T(r, 0), T(r, 1), T(r, 2), T(r, 3));
    if (good) std::printf("entropy %.6f\n", registerer.calcEntropy());
    std::printf("area_matched %.6f\n", registerer.getAreaMatched());
    return 0;
}
