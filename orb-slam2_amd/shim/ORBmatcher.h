// Drop-in replacement of the reference's include/ORBmatcher.h: the same public class — constructor, the static
// DescriptorDistance, the thirteen search methods and the three thresholds (include/ORBmatcher.h:39-97) — so that
// Tracking, LocalMapping, LoopClosing and MapPoint compile and call it unchanged.  The bodies (ORBmatcher.cc) read
// what a method needs out of the Frame / KeyFrame / MapPoint objects into flat arrays, hand them to the C ABI of
// include/orbhip.h, where projection, window search, Hamming distances, acceptance rules and the rotation filter run
// on the GPU, and write the results back into the caller's containers.
//
// Build inside the reference tree (its Frame.h, KeyFrame.h, MapPoint.h and DBoW2 headers on the include path), or,
// where that tree cannot be built, with -DORBHIP_SHIM_CV_STUB -Istandin against the stand-ins of shim/standin.
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

#include <climits>
#include <set>
#include <utility>
#include <vector>
#ifdef ORBHIP_SHIM_CV_STUB
#include "cv_stub.h"
#else
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#endif

#include "Frame.h"

struct orbhip_matcher;

namespace ORB_SLAM {

class KeyFrame;
class MapPoint;

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);

    // Hamming distance of two descriptor rows (host popcount: one pair is never worth a launch)
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);

    // Tracking: local map points with the projection Frame::isInFrustum left in them
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3);
    // Tracking: map points of the last frame projected with the current pose
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, float th);
    // Relocalisation: map points of a keyframe projected with the current pose
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, float th, int ORBdist);
    // Loop closing: map points projected with a similarity
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched,
                           int th);

    // Matching restricted to features under the same vocabulary node
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12);

    // Map points of frame 1 searched in frame 2 in a window around their position in frame 1
    int WindowSearch(Frame& F1, Frame& F2, int windowSize, std::vector<MapPoint*>& vpMapPointMatches2, int minOctave = -1,
                     int maxOctave = INT_MAX);
    // ... and around their projection with frame 2's pose
    int SearchByProjection(Frame& F1, Frame& F2, int windowSize, std::vector<MapPoint*>& vpMapPointMatches2);

    // Map initialisation
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                                int windowSize = 10);

    // Triangulation of new map points: unmatched features paired under the epipolar constraint
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<cv::KeyPoint>& vMatchedKeys1,
                               std::vector<cv::KeyPoint>& vMatchedKeys2, std::vector<std::pair<size_t, size_t> >& vMatchedPairs);

    // Map points of two keyframes matched through a similarity [s12*R12|t12]
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12,
                     const cv::Mat& t12, float th);

    // Duplicated map points: projected into a keyframe with its pose, or with a similarity
    int Fuse(KeyFrame* pKF, std::vector<MapPoint*>& vpMapPoints, float th = 2.5);
    int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th = 2.5);

public:
    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

    // The GPU handle the calling thread's matchers share (created on the thread's first search, destroyed when the
    // thread ends; ORBHIP_DEVICE selects the GPU).  Matchers are short-lived stack objects in the reference
    // (src/Tracking.cc:649, :717, :880 ...), so the handle does not live in them.  Also used by the forwarding bodies
    // of the steps around the matchers (FrameSteps.h).
    static orbhip_matcher* Accel();

protected:
    float mfNNratio;
    bool mbCheckOrientation;
};

}  // namespace ORB_SLAM

#endif  // ORBMATCHER_H
