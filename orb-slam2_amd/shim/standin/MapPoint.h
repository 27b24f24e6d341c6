// Test stand-in for ORB_SLAM::MapPoint (include/MapPoint.h:63-120): the accessors and tracking fields the
// matchers use, under the reference's names, over plain members a test can set.  Observations follow
// src/MapPoint.cc:72-85 and Replace follows :130-164 without the counters, the descriptor update and the map.
#ifndef ORBHIP_STANDIN_MAPPOINT_H
#define ORBHIP_STANDIN_MAPPOINT_H
#include <cstddef>
#include <map>

#include "Frame.h"

namespace ORB_SLAM {
class KeyFrame;

class MapPoint {
public:
    MapPoint()
        : mTrackProjX(0.f), mTrackProjY(0.f), mbTrackInView(false), mnTrackScaleLevel(0), mTrackViewCos(1.f), mbBad(false),
          mfMinDistance(0.f), mfMaxDistance(0.f) {}
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    cv::Mat GetNormal() { return mNormalVector.clone(); }
    void AddObservation(KeyFrame* pKF, size_t idx) {
        if (mObservations.count(pKF)) return;
        mObservations[pKF] = idx;
    }
    int GetIndexInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) ? (int)mObservations[pKF] : -1; }
    bool IsInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) != 0; }
    bool isBad() { return mbBad; }
    inline void Replace(MapPoint* pMP);  // defined in KeyFrame.h (it walks the observing keyframes)
    cv::Mat GetDescriptor() { return mDescriptor; }
    float GetMinDistanceInvariance() { return mfMinDistance; }
    float GetMaxDistanceInvariance() { return mfMaxDistance; }

    // written by Frame::isInFrustum, read by the local-map matcher
    float mTrackProjX, mTrackProjY;
    bool mbTrackInView;
    int mnTrackScaleLevel;
    float mTrackViewCos;

    // protected in the reference; public here so that a test can build a scene
    cv::Mat mWorldPos, mNormalVector, mDescriptor;
    std::map<KeyFrame*, size_t> mObservations;
    bool mbBad;
    float mfMinDistance, mfMaxDistance;
};
}  // namespace ORB_SLAM
#endif
