// register_pair_planes.cpp -- the plane side of Registration/RegisterPairRGBD360.cpp:60-110 of the reference on its own file
// formats: two `sphere_images_%d.bin` frames (Frame360::loadFrame) and the rig extrinsics `Rt_0N.txt` (Calib360::loadExtrinsicCalibration,
// Calib360.h:122-131: plain 4x4 text) -> per sensor: pinhole cloud down-sampled by 2 (Frame360.h:479-481), bilateral filter
// (Frame360.h:493-499), normal map + planar regions (Frame360.h:949-996), planes moved into the rig frame (Frame360.h:1046),
// co-planar pieces of several sensors merged (Frame360.h:655-733)
// -> RegisterRGBD360::RegisterPbMap(frame1, frame2, 25, PLANAR_3DoF) (RegisterPairRGBD360.cpp:101).
// Everything per-pixel runs on the GPU through the C ABI; the matcher and the pose fit are host code inside the library.
//   usage: register_pair_planes <frame1.bin> <frame2.bin> <extrinsics_dir> [regist_mode 0..3 = 1]
//   e.g.   register_pair_planes samples/sphere_images_1.bin samples/sphere_images_10.bin Calibration/Extrinsics
#include <array>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

#include "rgbd360/RegisterRGBD360.hpp"

static bool load_rt(const std::string& dir, int sensor, float Rt_colmajor[16]) {
    char name[1024];
    std::snprintf(name, sizeof(name), "%s/Rt_0%d.txt", dir.c_str(), sensor + 1);
    FILE* f = std::fopen(name, "r");
    if (!f) return false;
    bool ok = true;
    for (int r = 0; r < 4 && ok; ++r)
        for (int c = 0; c < 4 && ok; ++c) ok = std::fscanf(f, "%f", &Rt_colmajor[c * 4 + r]) == 1;
    std::fclose(f);
    return ok;
}

// the eight sensors of a frame are independent and each chain is launch-latency bound: one context (stream + device buffers)
// and one host thread per sensor, like the reference's `#pragma omp parallel num_threads(8)` (Frame360.h:489-503)
static bool frame_planes(std::array<rgbd360::RegisterPhotoICP, 8>& regs, const char* path, const std::string& extr,
                         std::vector<rgbd360_plane>& planes, size_t& n_pieces) {
    int rows = 0, cols = 0;
    if (rgbd360_load_frame_bin(path, nullptr, nullptr, &rows, &cols) != 0) return false;
    std::vector<uint8_t> rgb((size_t)8 * rows * cols * 3);
    std::vector<uint16_t> depth((size_t)8 * rows * cols);
    if (rgbd360_load_frame_bin(path, rgb.data(), depth.data(), &rows, &cols) != 0) return false;
    const int step = 2;                                                      // DOWNSAMPLE_160, Frame360.h:41
    rgbd360::SensorSegmentParams sp;
    sp.min_inliers = 40;                                                      // Frame360.h:960 is for the full 320 x 240 cloud: a quarter of the points
    std::array<std::vector<rgbd360_plane>, 8> per_sensor;
    std::array<int, 8> rc{};
    const auto tw0 = std::chrono::steady_clock::now();
    std::vector<std::thread> workers;
    for (int s = 0; s < 8; ++s)
        workers.emplace_back([&, s]() {
            float Rt[16];
            int n = 0;
            per_sensor[s].resize((size_t)sp.max_planes);
            rc[s] = load_rt(extr, s, Rt) ? 0 : 1;                             // depth image in, planes in the rig frame out: one call per sensor
            if (rc[s] == 0) rc[s] = rgbd360_set_plane_refinement(regs[s].context(), sp.refine ? 1 : 0, sp.refine_distance);      // segmentAndRefine, Frame360.h:977
            if (rc[s] == 0)
                rc[s] = rgbd360_sensor_planes(regs[s].context(), depth.data() + (size_t)s * rows * cols, (size_t)cols * 2, rows, cols, step, 0.3f,
                                              10.f, sp.sigma_s, sp.sigma_r, sp.max_depth_change_factor, sp.normal_smoothing_size, sp.min_inliers,
                                              sp.angular_threshold, sp.distance_threshold, sp.max_curvature, Rt, per_sensor[s].data(),
                                              sp.max_planes, &n);
            per_sensor[s].resize((size_t)(rc[s] == 0 ? n : 0));
        });
    for (std::thread& w : workers) w.join();
    if (std::getenv("RGBD360_EXAMPLE_TIMING"))
        std::fprintf(stderr, "  8 sensors on 8 threads: %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count());
    for (int s = 0; s < 8; ++s)
        if (rc[s] != 0) return false;
    n_pieces = 0;
    for (int s = 0; s < 8; ++s) n_pieces += per_sensor[s].size();
    planes = rgbd360::groupPlanes(std::vector<std::vector<rgbd360_plane>>(per_sensor.begin(), per_sensor.end()));      // Frame360::groupPlanes, Frame360.h:741-833
    return true;
}

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s frame1.bin frame2.bin extrinsics_dir [regist_mode]\n", argv[0]);
        return 2;
    }
    const int mode = argc > 4 ? std::atoi(argv[4]) : 1;
    std::array<rgbd360::RegisterPhotoICP, 8> regs;
    std::vector<rgbd360_plane> p1, p2;
    size_t n1 = 0, n2 = 0;
    if (!frame_planes(regs, argv[1], argv[3], p1, n1)) return 3;
    const auto t0 = std::chrono::steady_clock::now();                        // (the second frame: contexts and buffers exist)
    if (!frame_planes(regs, argv[2], argv[3], p2, n2)) return 3;
    std::fprintf(stderr, "planes of one frame (file read + 8 sensors on 8 threads + groupPlanes): %.3f ms\n",
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    p1 = rgbd360::mergePlanes(p1);                                            // Frame360::mergePlanes, Frame360.h:655-733
    p2 = rgbd360::mergePlanes(p2);
    std::printf("planes %zu %zu (pieces %zu %zu)\n", p1.size(), p2.size(), n1, n2);
    rgbd360::RegisterRGBD360 registerer(/*odometry_config=*/false);
    rgbd360::PlaneList f1{p1.data(), (int)p1.size()}, f2{p2.data(), (int)p2.size()};
    const bool good = registerer.RegisterPbMap(&f1, &f2, 25, (rgbd360::RegisterRGBD360::registrationType)mode);
    std::printf("status %d good %d matched %zu area_matched %.3f\n", registerer.status(), good ? 1 : 0, registerer.getMatchedPlanes().size(),
                registerer.getAreaMatched());
    const rgbd360::Mat4f T = registerer.getPose();
    for (int r = 0; r < 4; ++r) std::printf("%.6f %.6f %.6f %.6f\n", T(r, 0), T(r, 1), T(r, 2), T(r, 3));
    return 0;
}
