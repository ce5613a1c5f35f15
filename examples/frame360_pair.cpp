// The call sequence of the reference's Registration/RegisterPairRGBD360.cpp:60-90 through the Frame360 / Calib360 / RegisterRGBD360
// adapters (include/rgbd360/Frame360.hpp): two binary 8-sensor frames -> planes of every sensor on the device (getPlanes = the eight
// getPlanesSensor calls, groupPlanes, mergePlanes) -> RegisterPbMap; then, where the reference runs PCL's GICP on the sphere clouds
// (:112-142, third-party), the dense spherical alignment this library is built around, seeded with the plane pose
// (OdometryRGBD360.cpp:176-193's use of the same objects).  With an intrinsics directory the frames are undistorted first (:69, :76).
//   g++ -std=c++17 -O2 -pthread -Iinclude examples/frame360_pair.cpp -Lrgbd360_amd/lib -lrgbd360_hip -o frame360_pair
//   ./frame360_pair sphere_images_1.bin sphere_images_2.bin Calibration/Extrinsics [regist_mode [Calibration/Intrinsics]]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>

#include "rgbd360/Frame360.hpp"

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s frame1.bin frame2.bin extrinsics_dir [regist_mode [intrinsics_dir]]\n", argv[0]);
        return 2;
    }
    using namespace rgbd360;
    const std::string file360_1 = argv[1], file360_2 = argv[2];
    const int mode = argc > 4 ? std::atoi(argv[4]) : (int)RegisterRGBD360::PLANAR_3DoF;

    Calib360 calib;
    if (!calib.loadExtrinsicCalibration(argv[3])) return 3;
    const bool intrinsics = argc > 5 && calib.loadIntrinsicCalibration(argv[5]);      // Calibration/Intrinsics: frames are then undistorted like the source's

    try {
        Frame360 frame360_1(&calib);
        frame360_1.loadFrame(file360_1);
        if (intrinsics) frame360_1.undistort();               // RegisterPairRGBD360.cpp:69
        frame360_1.fastStitchImage360();                      // (the viewer's quick panorama; replaced by the spherical one below)
        {
            const int R = frame360_1.sensorRows(), Cn = frame360_1.sensorCols();
            const uint8_t* pano = (const uint8_t*)frame360_1.sphereRGB.data;
            const uint8_t* s7 = (const uint8_t*)frame360_1.sensorRGB(7).data;
            // strip 0 holds sensor 7 transposed and flipped: panorama (r, c) = sensor (c, cols - 1 - r)
            bool same = frame360_1.sphereRGB.rows == Cn && frame360_1.sphereRGB.cols == 8 * R;
            for (int r = 0; r < Cn && same; r += 37)
                for (int c = 0; c < R && same; c += 41)
                    for (int ch = 0; ch < 3; ++ch) same = same && pano[((size_t)r * 8 * R + c) * 3 + ch] == s7[((size_t)c * Cn + (Cn - 1 - r)) * 3 + ch];
            if (!same) { std::fprintf(stderr, "fastStitchImage360: unexpected panorama\n"); return 4; }
        }
        frame360_1.stitchSphericalImage();
        frame360_1.buildSphereCloud();                        // RegisterPairRGBD360.cpp:70 (the rig-frame cloud of the eight sensors)
        frame360_1.getPlanes();

        Frame360 frame360_2(&calib);
        frame360_2.loadFrame(file360_2);
        if (intrinsics) frame360_2.undistort();
        frame360_2.stitchSphericalImage();
        const auto t0 = std::chrono::steady_clock::now();
        frame360_2.getPlanes();
        std::fprintf(stderr, "getPlanes: %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        size_t pieces1 = 0, pieces2 = 0, coloured = 0;
        for (int s = 0; s < 8; ++s) {
            pieces1 += frame360_1.local_planes_[(size_t)s].vPlanes.size();
            pieces2 += frame360_2.local_planes_[(size_t)s].vPlanes.size();
            // Frame360.h:1034, :1041, :1045-1046: nothing smaller than min_area_plane or narrower than max_elongation_plane is stored,
            // and every stored plane carries its colour descriptors
            for (const rgbd360_plane& p : frame360_1.local_planes_[(size_t)s].vPlanes) {
                if (p.area < 0.12f || p.elongation > 6.f) { std::fprintf(stderr, "sensor %d stores a plane of %.3f m2, elongation %.1f\n", s, p.area, p.elongation); return 4; }
                coloured += p.color_count > 0;
            }
        }
        if (coloured != pieces1) { std::fprintf(stderr, "%zu of %zu sensor planes carry colour\n", coloured, pieces1); return 4; }
        {
            // every plane of the frame lies in the rig frame like the cloud: the points of sensor clouds must sit on some plane of the list
            size_t finite = 0, on_plane = 0;
            const std::vector<float>& c = frame360_1.sphereCloud;
            for (size_t i = 0; i + 2 < c.size(); i += 3 * 97) {
                if (!(c[i] == c[i])) continue;
                ++finite;
                for (const rgbd360_plane& p : frame360_1.planes.vPlanes)
                    if (std::fabs(p.normal[0] * c[i] + p.normal[1] * c[i + 1] + p.normal[2] * c[i + 2] + p.d) < 0.03f) { ++on_plane; break; }
            }
            std::printf("sphere cloud %zu points, sampled %zu finite, %zu within 3 cm of a plane of the frame\n", c.size() / 3, finite, on_plane);
        }
        std::printf("planes %zu %zu (pieces %zu %zu) planar area %.3f %.3f average intensity %d %d\n", frame360_1.planes.vPlanes.size(),
                    frame360_2.planes.vPlanes.size(), pieces1, pieces2, frame360_1.getPlanarArea(), frame360_2.getPlanarArea(),
                    frame360_1.getAverageIntensity(), frame360_2.getAverageIntensity());

        RegisterRGBD360 registerer(/*odometry_config=*/true);      // configLocaliser_sphericalOdometry.ini
        const bool good = registerer.RegisterPbMap(&frame360_1, &frame360_2, 25, (RegisterRGBD360::registrationType)mode);
        std::map<unsigned, unsigned> bestMatch = registerer.getMatchedPlanes();
        std::printf("status %d good %d matched %zu\n", registerer.status(), good ? 1 : 0, bestMatch.size());
        const Mat4f Tp = good ? registerer.getPose() : Mat4f::Identity();
        for (int r = 0; r < 4; ++r) std::printf("%.6f %.6f %.6f %.6f\n", Tp(r, 0), Tp(r, 1), Tp(r, 2), Tp(r, 3));

        // the dense alignment of the two panoramas, from the plane pose
        RegisterPhotoICP align360;
        align360.setNumPyr(3);
        align360.setTargetFrame(frame360_1.sphereRGB, frame360_1.sphereDepth);
        align360.setSourceFrame(frame360_2.sphereRGB, frame360_2.sphereDepth);
        align360.alignFrames360(Tp, RegisterPhotoICP::PHOTO_DEPTH);
        const Mat4f Td = align360.getOptimalPosePod();
        std::printf("dense status %d\n", align360.status());
        for (int r = 0; r < 4; ++r) std::printf("%.6f %.6f %.6f %.6f\n", Td(r, 0), Td(r, 1), Td(r, 2), Td(r, 3));
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 3;
    }
    return 0;
}
