// The accelerated primitive of the reference's ORBmatcher, with its signature unchanged:
//   static int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b)
// (include/ORBmatcher.h:46, src/ORBmatcher.cc:1805-1821).  In the reference tree the body of that
// one function is replaced by the single forwarding line below; the 13 Search*/Fuse methods keep
// their CPU bodies and call it as before.  Batched dense matching goes through orbhip_knn2*.
#ifndef ORBMATCHER_ACCEL_H
#define ORBMATCHER_ACCEL_H
#ifdef ORBHIP_SHIM_CV_STUB
#include "cv_stub.h"
#else
#include <opencv2/core/core.hpp>
#endif
#include "../../include/orbhip.h"

namespace ORB_SLAM {
struct ORBmatcherAccel {
    static const int TH_HIGH = 100;
    static const int TH_LOW = 20;
    static const int HISTO_LENGTH = 30;
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {
        return orbhip_hamming_distance(a.ptr<int32_t>(), b.ptr<int32_t>());
    }
};
}  // namespace ORB_SLAM
#endif
