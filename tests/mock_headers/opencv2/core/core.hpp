// Minimal stand-in for <opencv2/core/core.hpp> -- TEST INFRASTRUCTURE ONLY (tests/test_cpp_adapter.py): the handful of cv::Mat
// members include/rgbd360/RegisterPhotoICP.hpp touches (data, rows, cols, step, type(), create()).  Not shipped.
#pragma once
#include <cstddef>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_16U 2
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_16UC1 CV_MAKETYPE(CV_16U, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)

namespace cv {

class Mat {
   public:
    unsigned char* data = nullptr;
    int rows = 0, cols = 0;
    size_t step = 0;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    int type() const { return type_; }
    void create(int r, int c, int type) {
        static const size_t depth_bytes[8] = {1, 1, 2, 2, 4, 4, 8, 0};
        const size_t elem = depth_bytes[type & 7] * (size_t)((type >> 3) + 1);
        rows = r; cols = c; type_ = type; step = (size_t)c * elem;
        store_ = std::make_shared<std::vector<unsigned char>>((size_t)r * step);
        data = store_->data();
    }

   private:
    int type_ = 0;
    std::shared_ptr<std::vector<unsigned char>> store_;
};

}  // namespace cv
