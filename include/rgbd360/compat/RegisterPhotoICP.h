// Drop-in for the reference's `#include <RegisterPhotoICP.h>`: the MI355X-backed class under the reference's header name and in
// the global namespace, where the reference's applications expect it (OdometryRGBD360.cpp:36, LoopClosure360.h:39,
// KFsphere_SLAM.cpp:32 all say `RegisterPhotoICP align360;`).  Put include/rgbd360/compat in front of the reference's include
// directory, or change the include line; nothing else at the call sites changes.  With Eigen and OpenCV on the include path the
// class has the reference's signatures (cv::Mat frames, Eigen poses / Hessians): see ../RegisterPhotoICP.hpp.
#pragma once
#include "../RegisterPhotoICP.hpp"
using rgbd360::RegisterPhotoICP;
