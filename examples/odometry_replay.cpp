// odometry_replay.cpp -- replays the call sequence of Registration/OdometryRGBD360.cpp:141-297 of the reference
// (per frame: set target / set source / alignFrames360(PHOTO_DEPTH) / accumulate currentPose) through the C++
// adapter, on frames read from raw files written by tools/dump_sequence.py:
//     frame_%03d.rgb  (H*W*3 uint8)   frame_%03d.depth (H*W uint16 mm)
// Build:  g++ -std=c++17 -O2 -Iinclude examples/odometry_replay.cpp -Lrgbd360_amd/lib -lrgbd360_hip
//             -Wl,-rpath,$PWD/rgbd360_amd/lib -o odometry_replay
// Usage:  odometry_replay <dir> <n_frames> <width> <height> [--sequence | --multi <n_gpus> | --pbmap | --link]
//         --sequence: all frames are loaded first and the frame loop runs inside the library (alignSequence)
//         --multi N:  the same sequence sharded over N GPUs of this node from this one process (rgbd360_multi_*: one host thread
//                     per device, contiguous shards of pairs, one ncclAllGather of the solved poses over xGMI); prints the
//                     lines of --sequence
//         --pbmap:    every pair is first registered from its planes (RegisterRGBD360::RegisterPbMap, ODOMETRY_6DoF, as
//                     SphereGraphSLAM.cpp:180 / KFsphere_SLAM.cpp:314 do) and that pose seeds alignFrames360
//                     (KFsphere_SLAM.cpp:149); prints one extra "pbmap" line per pair
//         --link:     the same per pair through the one-call form rgbd360::RegisterFrames (planes, RegisterPbMap, seeded dense
//                     alignment, the reference's isApprox(1e-1) validity test); prints "link <pair> <ok> rel_t ..."
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "rgbd360/RegisterPhotoICP.hpp"
#include "rgbd360/RegisterRGBD360.hpp"

struct Frame {   // the two members of Frame360 the dense path reads (Frame360.h:104-111)
    std::vector<uint8_t> rgb;
    std::vector<uint16_t> depth;
    int rows, cols;
    rgbd360::ImageView sphereRGB, sphereDepth;
    bool load(const std::string& dir, int k, int w, int h) {
        rows = h; cols = w;
        char name[512];
        rgb.resize((size_t)w * h * 3);
        depth.resize((size_t)w * h);
        snprintf(name, sizeof(name), "%s/frame_%03d.rgb", dir.c_str(), k);
        std::ifstream f1(name, std::ios::binary);
        if (!f1.read((char*)rgb.data(), rgb.size())) return false;
        snprintf(name, sizeof(name), "%s/frame_%03d.depth", dir.c_str(), k);
        std::ifstream f2(name, std::ios::binary);
        if (!f2.read((char*)depth.data(), depth.size() * 2)) return false;
        sphereRGB = {rgb.data(), h, w, (size_t)w * 3, rgbd360::ImageView::U8C3};
        sphereDepth = {depth.data(), h, w, (size_t)w * 2, rgbd360::ImageView::U16C1};
        return true;
    }
};

static rgbd360::Mat4f mul(const rgbd360::Mat4f& A, const rgbd360::Mat4f& B) {
    rgbd360::Mat4f C{};
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float s = 0;
            for (int k = 0; k < 4; ++k) s += A(r, k) * B(k, c);
            C(r, c) = s;
        }
    return C;
}

int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s <dir> <n_frames> <width> <height> [--sequence | --pbmap]\n", argv[0]);
        return 2;
    }
    const std::string dir = argv[1];
    const int n = atoi(argv[2]), w = atoi(argv[3]), h = atoi(argv[4]);
    rgbd360::RegisterPhotoICP align360;      // OdometryRGBD360.cpp:91-95
    align360.setNumPyr(4);
    align360.useSaliency(false);
    rgbd360::Mat4f currentPose = rgbd360::Mat4f::Identity();
    if (argc > 5 && std::string(argv[5]) == "--sequence") {
        std::vector<Frame> frames(n);
        std::vector<rgbd360::ImageView> rgb, depth;
        for (int k = 0; k < n; ++k) {
            if (!frames[k].load(dir, k, w, h)) return 3;
            rgb.push_back(frames[k].sphereRGB);
            depth.push_back(frames[k].sphereDepth);
        }
        std::vector<rgbd360_result> res;
        const std::vector<rgbd360::Mat4f> rels = align360.alignSequence(rgb, depth, rgbd360::RegisterPhotoICP::PHOTO_DEPTH, 0, 3,
                                                                        rgbd360::Mat4f::Identity(), &res);
        for (size_t j = 0; j < rels.size(); ++j) {
            currentPose = mul(currentPose, rels[j]);
            printf("pair %zu status %d sso %.4f rel_t %.5f %.5f %.5f pose_t %.5f %.5f %.5f\n", j, res[j].status, res[j].sso,
                   rels[j](0, 3), rels[j](1, 3), rels[j](2, 3), currentPose(0, 3), currentPose(1, 3), currentPose(2, 3));
        }
        return 0;
    }
    if (argc > 6 && std::string(argv[5]) == "--multi") {
        const int n_gpus = atoi(argv[6]);
        std::vector<Frame> frames(n);
        std::vector<const uint8_t*> rgb(n);
        std::vector<const void*> depth(n);
        for (int k = 0; k < n; ++k) {
            if (!frames[k].load(dir, k, w, h)) return 3;
            rgb[k] = frames[k].rgb.data();
            depth[k] = frames[k].depth.data();
        }
        rgbd360_params p;
        rgbd360_default_params(&p);
        p.n_pyr = 4;
        rgbd360_multi* m = nullptr;
        int rc = rgbd360_multi_create(&p, n_gpus, nullptr, &m);
        if (rc != 0) {
            fprintf(stderr, "rgbd360_multi_create(%d GPUs) failed: %d\n", n_gpus, rc);
            return 4;
        }
        std::vector<float> poses((size_t)(n - 1) * 16);
        std::vector<rgbd360_result> res(n - 1);
        rc = rgbd360_multi_align_sequence(m, n, rgb.data(), (size_t)w * 3, depth.data(), (size_t)w * 2, 0, h, w, nullptr, 2, 0, 3, poses.data(),
                                          res.data());
        if (rc != 0) {
            fprintf(stderr, "rgbd360_multi_align_sequence: %s (%d)\n", rgbd360_multi_last_error(m), rc);
            rgbd360_multi_destroy(m);
            return 5;
        }
        for (int j = 0; j + 1 < n; ++j) {
            rgbd360::Mat4f rel{};
            for (int k = 0; k < 16; ++k) rel.m[k] = poses[(size_t)j * 16 + k];
            currentPose = mul(currentPose, rel);                                                // OdometryRGBD360.cpp:257
            printf("pair %d status %d sso %.4f rel_t %.5f %.5f %.5f pose_t %.5f %.5f %.5f\n", j, res[j].status, res[j].sso, rel(0, 3), rel(1, 3),
                   rel(2, 3), currentPose(0, 3), currentPose(1, 3), currentPose(2, 3));
        }
        rgbd360_multi_destroy(m);
        return 0;
    }
    if (argc > 5 && std::string(argv[5]) == "--link") {
        rgbd360::RegisterRGBD360 registerer(/*odometry_config=*/true);
        rgbd360::SegmentParams seg;
        seg.max_depth_change_factor = 0.05f;
        seg.min_inliers = 40;
        seg.angular_threshold = 0.03f;
        seg.distance_threshold = 0.05f;
        Frame a, b;
        if (!a.load(dir, 0, w, h)) return 3;
        for (int k = 1; k < n; ++k) {
            if (!b.load(dir, k, w, h)) return 3;
            rgbd360::Mat4f rel = rgbd360::Mat4f::Identity();
            const bool ok = rgbd360::RegisterFrames(a, b, rel, [](const rgbd360::ImageView& v) { return v; }, align360, registerer,
                                                    rgbd360::RegisterRGBD360::ODOMETRY_6DoF, 25, seg);
            printf("link %d ok %d matched %zu rel_t %.5f %.5f %.5f\n", k - 1, ok ? 1 : 0, registerer.getMatchedPlanes().size(), rel(0, 3),
                   rel(1, 3), rel(2, 3));
            std::swap(a, b);
        }
        return 0;
    }
    const bool use_pbmap = argc > 5 && std::string(argv[5]) == "--pbmap";
    rgbd360::RegisterRGBD360 registerer(/*odometry_config=*/true);
    rgbd360::SegmentParams seg;
    seg.max_depth_change_factor = 0.05f;        // the synthetic frames are full spheres: Frame360_stereo.h:854-882 set-up
    seg.min_inliers = 40;
    seg.angular_threshold = 0.03f;
    seg.distance_threshold = 0.05f;
    std::vector<rgbd360_plane> planes1, planes2;
    Frame frame1, frame2;
    if (!frame1.load(dir, 0, w, h)) return 3;
    if (use_pbmap) planes1 = rgbd360::segmentPlanes(align360, frame1.sphereDepth, seg);
    for (int k = 1; k < n; ++k) {
        if (!frame2.load(dir, k, w, h)) return 3;
        rgbd360::Mat4f guess = rgbd360::Mat4f::Identity();
        if (use_pbmap) {
            planes2 = rgbd360::segmentPlanes(align360, frame2.sphereDepth, seg);
            rgbd360::PlaneList ref{planes1.data(), (int)planes1.size()}, trg{planes2.data(), (int)planes2.size()};
            const bool good = registerer.RegisterPbMap(&ref, &trg, 25, rgbd360::RegisterRGBD360::ODOMETRY_6DoF);
            if (good) guess = registerer.getPose();
            printf("pbmap %d good %d matched %zu t %.5f %.5f %.5f\n", k - 1, good ? 1 : 0, registerer.getMatchedPlanes().size(), guess(0, 3),
                   guess(1, 3), guess(2, 3));
        }
        if (k == 1) align360.setTargetFrame(frame1.sphereRGB, frame1.sphereDepth);              // :189
        else align360.promoteSourceToTarget();          // frame1 is last step's frame2: already on the device
        align360.setSourceFrame(frame2.sphereRGB, frame2.sphereDepth);                          // :190
        align360.alignFrames360(guess, rgbd360::RegisterPhotoICP::PHOTO_DEPTH);                 // :192
        const rgbd360::Mat4f rel = align360.getOptimalPosePod();                                  // :193
        currentPose = mul(currentPose, rel);                                                    // :257
        printf("pair %d status %d sso %.4f rel_t %.5f %.5f %.5f pose_t %.5f %.5f %.5f\n", k - 1, align360.status(), align360.SSO,
               rel(0, 3), rel(1, 3), rel(2, 3), currentPose(0, 3), currentPose(1, 3), currentPose(2, 3));
        fprintf(stderr, "entropy %d %.5f\n", k - 1, align360.calcEntropy());                     // :207 (commented out in the source)
        std::swap(frame1, frame2);
        std::swap(planes1, planes2);
    }
    return 0;
}
