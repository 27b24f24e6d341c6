// Test stand-in for ORB_SLAM::Frame (include/Frame.h:44-150): the public data members the ORB front end reads
// and writes, under the reference's names, and nothing else.  The drop-in ORBmatcher is compiled against this
// header where the reference tree (OpenCV 2.4, ROS, boost) is not available; in a real tree the reference's own
// Frame.h is found first and this file is not used.
#ifndef ORBHIP_STANDIN_FRAME_H
#define ORBHIP_STANDIN_FRAME_H
#include <cstddef>
#include <vector>

#ifdef ORBHIP_SHIM_CV_STUB
#include "../cv_stub.h"
#else
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#endif
#include "DBoW2.h"

namespace ORB_SLAM {
class MapPoint;
class KeyFrame;

class Frame {
public:
    Frame() : mnScaleLevels(0), mfScaleFactor(1.f) {}
    cv::Mat mK;
    static float fx, fy, cx, cy;
    cv::Mat mDistCoef;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    DBoW2::BowVector mBowVec;
    DBoW2::FeatureVector mFeatVec;
    cv::Mat mDescriptors;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    cv::Mat mTcw;
    int mnScaleLevels;
    float mfScaleFactor;
    std::vector<float> mvScaleFactors;
    static int mnMinX, mnMaxX, mnMinY, mnMaxY;
};
}  // namespace ORB_SLAM
#endif
