// Test stand-in for ORB_SLAM::KeyFrame (include/KeyFrame.h:58-160): the accessors the matchers use, under the
// reference's names, over plain members a test can set.  SetPose keeps Rcw, tcw and Ow = -Rcw.t()*tcw as
// src/KeyFrame.cc:102-109 does; IsInImage is the half-open test of :680-683 on the bounds every keyframe copies
// from Frame (:44).
#ifndef ORBHIP_STANDIN_KEYFRAME_H
#define ORBHIP_STANDIN_KEYFRAME_H
#include <cstddef>
#include <set>
#include <vector>

#include "Frame.h"
#include "MapPoint.h"

namespace ORB_SLAM {

class KeyFrame {
public:
    KeyFrame() : fx(0.f), fy(0.f), cx(0.f), cy(0.f) {}
    void SetPose(const cv::Mat& Tcw_) {
        mRcw = Tcw_.rowRange(0, 3).colRange(0, 3).clone();
        mtcw = Tcw_.rowRange(0, 3).col(3).clone();
        Ow = -mRcw.t() * mtcw;
    }
    cv::Mat GetCameraCenter() { return Ow.clone(); }
    cv::Mat GetRotation() { return mRcw.clone(); }
    cv::Mat GetTranslation() { return mtcw.clone(); }
    DBoW2::FeatureVector GetFeatureVector() { return mFeatVec; }
    void AddMapPoint(MapPoint* pMP, const size_t& idx) { mvpMapPoints[idx] = pMP; }
    void EraseMapPointMatch(const size_t& idx) { mvpMapPoints[idx] = 0; }
    void ReplaceMapPointMatch(const size_t& idx, MapPoint* pMP) { mvpMapPoints[idx] = pMP; }
    std::set<MapPoint*> GetMapPoints() {
        std::set<MapPoint*> s;
        for (size_t i = 0; i < mvpMapPoints.size(); i++)
            if (mvpMapPoints[i]) s.insert(mvpMapPoints[i]);
        return s;
    }
    std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
    MapPoint* GetMapPoint(const size_t& idx) { return mvpMapPoints[idx]; }
    cv::KeyPoint GetKeyPointUn(const size_t& idx) const { return mvKeysUn[idx]; }
    std::vector<cv::KeyPoint> GetKeyPointsUn() const { return mvKeysUn; }
    cv::Mat GetDescriptors() { return mDescriptors.clone(); }
    bool IsInImage(const float& x, const float& y) const {
        return x >= Frame::mnMinX && x < Frame::mnMaxX && y >= Frame::mnMinY && y < Frame::mnMaxY;
    }
    float GetScaleFactor(int nLevel = 1) const { return mvScaleFactors[nLevel]; }
    std::vector<float> GetScaleFactors() const { return mvScaleFactors; }
    float GetSigma2(int nLevel = 1) const { return mvLevelSigma2[nLevel]; }
    int GetScaleLevels() const { return (int)mvScaleFactors.size(); }

    float fx, fy, cx, cy;

    // protected in the reference; public here so that a test can build a scene
    cv::Mat mRcw, mtcw, Ow;
    std::vector<cv::KeyPoint> mvKeysUn;
    cv::Mat mDescriptors;
    std::vector<MapPoint*> mvpMapPoints;
    DBoW2::FeatureVector mFeatVec;
    std::vector<float> mvScaleFactors, mvLevelSigma2;
};

inline void MapPoint::Replace(MapPoint* pMP) {
    if (pMP == this) return;
    std::map<KeyFrame*, size_t> obs = mObservations;
    mObservations.clear();
    mbBad = true;
    for (std::map<KeyFrame*, size_t>::iterator mit = obs.begin(), mend = obs.end(); mit != mend; mit++) {
        KeyFrame* pKF = mit->first;
        if (!pMP->IsInKeyFrame(pKF)) {
            pKF->ReplaceMapPointMatch(mit->second, pMP);
            pMP->AddObservation(pKF, mit->second);
        } else {
            pKF->EraseMapPointMatch(mit->second);
        }
    }
}
}  // namespace ORB_SLAM
#endif
