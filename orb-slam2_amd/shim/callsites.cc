// Compile-time check of the drop-in claim: every call the reference's three threads make on an ORBmatcher, each
// spelled the way the reference spells it (constructor arguments, argument order, defaulted arguments left out,
// literals of the types the reference passes), against shim/ORBmatcher.h.  Cited per block; the surrounding
// control flow of the callers is not reproduced.  The functions are never run.
#include <set>
#include <vector>

#include "KeyFrame.h"
#include "MapPoint.h"
#include "ORBmatcher.h"

namespace ORB_SLAM {
using namespace std;

struct CallSites {
    Frame mCurrentFrame, mLastFrame, mInitialFrame;
    vector<cv::Point2f> mvbPrevMatched;
    vector<int> mvIniMatches;
    vector<MapPoint*> mvpLocalMapPoints, mvpLoopMapPoints, mvpCurrentMatchedPoints;
    KeyFrame *mpCurrentKeyFrame, *mpCurrentKF;
    cv::Mat mScw;

    // src/Tracking.cc:512-513
    int Initialize() {
        ORBmatcher matcher(0.9,true);
        int nmatches = matcher.SearchForInitialization(mInitialFrame,mCurrentFrame,mvbPrevMatched,mvIniMatches,100);
        return nmatches;
    }
    // src/Tracking.cc:649-663, :689, :692
    int TrackPreviousFrame(int minOctave) {
        ORBmatcher matcher(0.9,true);
        vector<MapPoint*> vpMapPointMatches;
        int nmatches = matcher.WindowSearch(mLastFrame,mCurrentFrame,200,vpMapPointMatches,minOctave);
        if(nmatches<10)
            nmatches = matcher.WindowSearch(mLastFrame,mCurrentFrame,100,vpMapPointMatches,0);
        nmatches += matcher.SearchByProjection(mLastFrame,mCurrentFrame,15,vpMapPointMatches);
        nmatches = matcher.SearchByProjection(mLastFrame,mCurrentFrame,50,vpMapPointMatches);
        return nmatches;
    }
    // src/Tracking.cc:717-726
    int TrackWithMotionModel() {
        ORBmatcher matcher(0.9,true);
        int nmatches = matcher.SearchByProjection(mCurrentFrame,mLastFrame,15);
        return nmatches;
    }
    // src/Tracking.cc:880-885
    void SearchReferencePointsInFrustum() {
        ORBmatcher matcher(0.8);
        int th = 1;
        matcher.SearchByProjection(mCurrentFrame,mvpLocalMapPoints,th);
    }
    // src/Tracking.cc:1050, :1070, :1091, :1156, :1170
    int Relocalisation(vector<KeyFrame*> &vpCandidateKFs, vector<vector<MapPoint*> > &vvpMapPointMatches, set<MapPoint*> &sFound, int i) {
        ORBmatcher matcher(0.75,true);
        KeyFrame* pKF = vpCandidateKFs[i];
        int nmatches = matcher.SearchByBoW(pKF,mCurrentFrame,vvpMapPointMatches[i]);
        ORBmatcher matcher2(0.9,true);
        int nadditional =matcher2.SearchByProjection(mCurrentFrame,vpCandidateKFs[i],sFound,10,100);
        nadditional =matcher2.SearchByProjection(mCurrentFrame,vpCandidateKFs[i],sFound,3,64);
        return nmatches + nadditional;
    }
    // src/LocalMapping.cc:219, :261
    void CreateNewMapPoints(KeyFrame* pKF2, cv::Mat F12) {
        ORBmatcher matcher(0.6,false);
        vector<cv::KeyPoint> vMatchedKeysUn1;
        vector<cv::KeyPoint> vMatchedKeysUn2;
        vector<pair<size_t,size_t> > vMatchedIndices;
        matcher.SearchForTriangulation(mpCurrentKeyFrame,pKF2,F12,vMatchedKeysUn1,vMatchedKeysUn2,vMatchedIndices);
    }
    // src/LocalMapping.cc:408-439
    void SearchInNeighbors(vector<KeyFrame*> &vpTargetKFs, vector<MapPoint*> &vpFuseCandidates) {
        ORBmatcher matcher(0.6);
        vector<MapPoint*> vpMapPointMatches = mpCurrentKeyFrame->GetMapPointMatches();
        for(vector<KeyFrame*>::iterator vit=vpTargetKFs.begin(), vend=vpTargetKFs.end(); vit!=vend; vit++)
        {
            KeyFrame* pKFi = *vit;
            matcher.Fuse(pKFi,vpMapPointMatches);
        }
        matcher.Fuse(mpCurrentKeyFrame,vpFuseCandidates);
    }
    // src/LoopClosing.cc:263, :289, :347, :400
    int ComputeSim3(KeyFrame* pKF, vector<vector<MapPoint*> > &vvpMapPointMatches, int i, float s, cv::Mat R, cv::Mat t) {
        ORBmatcher matcher(0.75,true);
        int nmatches = matcher.SearchByBoW(mpCurrentKF,pKF,vvpMapPointMatches[i]);
        vector<MapPoint*> vpMapPointMatches(vvpMapPointMatches[i].size(), static_cast<MapPoint*>(NULL));
        matcher.SearchBySim3(mpCurrentKF,pKF,vpMapPointMatches,s,R,t,7.5);
        matcher.SearchByProjection(mpCurrentKF, mScw, mvpLoopMapPoints, mvpCurrentMatchedPoints,10);
        return nmatches;
    }
    // src/LoopClosing.cc:589-598
    void SearchAndFuse(KeyFrame* pKF, cv::Mat cvScw) {
        ORBmatcher matcher(0.8);
        matcher.Fuse(pKF,cvScw,mvpLoopMapPoints,4);
    }
    // src/MapPoint.cc:230
    int ComputeDistinctiveDescriptors(vector<cv::Mat> &vDescriptors, size_t i, size_t j) {
        int distij = ORBmatcher::DescriptorDistance(vDescriptors[i],vDescriptors[j]);
        return distij + ORBmatcher::TH_LOW + ORBmatcher::TH_HIGH + ORBmatcher::HISTO_LENGTH;
    }
};

// referenced so that the translation unit cannot be discarded as unused
void* orbhip_shim_callsites_anchor() {
    static int (CallSites::*p)() = &CallSites::TrackWithMotionModel;
    return (void*)&p;
}

}  // namespace ORB_SLAM
