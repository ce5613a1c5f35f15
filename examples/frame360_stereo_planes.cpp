// Frame360_stereo through its adapter (include/rgbd360/Frame360.hpp): the raw range panorama of the stereo omnidirectional camera ->
// sphere cloud -> planes (Frame360_stereo.h:268-311, 454-512, 847-980).
//   g++ -std=c++17 -O2 -Iinclude examples/frame360_stereo_planes.cpp -Lrgbd360_amd/lib -lrgbd360_hip -o frame360_stereo_planes
//   ./frame360_stereo_planes depth.raw
#include <cmath>
#include <cstdio>
#include <string>

#include "rgbd360/Frame360.hpp"

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s depth.raw\n", argv[0]);
        return 2;
    }
    try {
        rgbd360::Frame360_stereo frame;
        frame.loadDepth(argv[1]);
        frame.buildSphereCloud();
        frame.getPlanesStereo();
        size_t finite = 0;
        for (size_t i = 0; i < frame.sphereCloud.size(); i += 3) finite += std::isfinite(frame.sphereCloud[i]) ? 1 : 0;
        std::printf("image %d x %d, cloud points %zu, planes %zu\n", frame.sphereDepth.rows, frame.sphereDepth.cols, finite, frame.planes.vPlanes.size());
        for (const rgbd360_plane& p : frame.planes.vPlanes)
            std::printf("%d %.4f %.4f %.4f %.4f %.3f\n", p.count, p.normal[0], p.normal[1], p.normal[2], p.d, p.area);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 3;
    }
    return 0;
}
