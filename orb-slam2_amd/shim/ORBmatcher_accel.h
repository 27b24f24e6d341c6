// The accelerated primitive of the reference's ORBmatcher, with its signature unchanged:
//   static int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b)
// (include/ORBmatcher.h:46, src/ORBmatcher.cc:1805-1821).  In the reference tree the body of that
// one function is replaced by the single forwarding line below; Search*/Fuse methods without a forwarding
// body here keep their CPU bodies and call it as before.  Batched dense matching goes through orbhip_knn2*.
#ifndef ORBMATCHER_ACCEL_H
#define ORBMATCHER_ACCEL_H
#ifdef ORBHIP_SHIM_CV_STUB
#include "cv_stub.h"
#else
#include <opencv2/core/core.hpp>
#endif
#include <climits>
#include <stdexcept>
#include <vector>

#include "../../include/orbhip.h"

namespace ORB_SLAM {
struct ORBmatcherAccel {
    static const int TH_HIGH = 100;
    static const int TH_LOW = 20;
    static const int HISTO_LENGTH = 30;
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {
        return orbhip_hamming_distance(a.ptr<int32_t>(), b.ptr<int32_t>());
    }

    // The two windowed matchers, with the reference's signatures (include/ORBmatcher.h:71,76).  In the
    // reference tree these bodies replace src/ORBmatcher.cc:420-527 and :609-724; FrameT is
    // ORB_SLAM::Frame (members mvKeysUn, mDescriptors, mvpMapPoints and the static image bounds
    // mnMinX..mnMaxY, include/Frame.h).  MapPointT needs isBad().
    float mfNNratio;
    bool mbCheckOrientation;
    orbhip_matcher* handle;
    explicit ORBmatcherAccel(float nnratio = 0.6, bool checkOri = true, int device = 0)
        : mfNNratio(nnratio), mbCheckOrientation(checkOri), handle(0) {
        if (orbhip_matcher_create(&handle, device) != ORBHIP_OK) throw std::runtime_error(orbhip_last_error());
    }
    ~ORBmatcherAccel() { orbhip_matcher_destroy(handle); }

    template <class FrameT>
    static std::vector<unsigned char> packDescriptors(const FrameT& F) {
        std::vector<unsigned char> d((size_t)F.mDescriptors.rows * 32);
        for (int i = 0; i < F.mDescriptors.rows; i++) {
            const unsigned char* r = F.mDescriptors.template ptr<unsigned char>(i);
            for (int b = 0; b < 32; b++) d[(size_t)i * 32 + b] = r[b];
        }
        return d;
    }

    template <class FrameT>
    int SearchForInitialization(FrameT& F1, FrameT& F2, std::vector<cv::Point2f>& vbPrevMatched,
                                std::vector<int>& vnMatches12, int windowSize = 10) {
        static_assert(sizeof(cv::KeyPoint) == sizeof(OrbKeypoint), "cv::KeyPoint layout");
        static_assert(sizeof(cv::Point2f) == 8, "cv::Point2f layout");
        const int n1 = (int)F1.mvKeysUn.size(), n2 = (int)F2.mvKeysUn.size();
        vnMatches12.assign(n1, -1);
        const std::vector<unsigned char> d1 = packDescriptors(F1), d2 = packDescriptors(F2);
        const OrbFrameBounds b = {FrameT::mnMinX, FrameT::mnMinY, FrameT::mnMaxX, FrameT::mnMaxY};
        int nmatches = 0;
        if (orbhip_search_for_initialization(handle, (const OrbKeypoint*)F1.mvKeysUn.data(), d1.data(), n1,
                                             (const OrbKeypoint*)F2.mvKeysUn.data(), d2.data(), n2, b,
                                             (float*)vbPrevMatched.data(), windowSize, mfNNratio, mbCheckOrientation,
                                             vnMatches12.data(), &nmatches) != ORBHIP_OK)
            throw std::runtime_error(orbhip_last_error());
        return nmatches;
    }

    template <class FrameT, class MapPointT>
    int WindowSearch(FrameT& F1, FrameT& F2, int windowSize, std::vector<MapPointT*>& vpMapPointMatches2,
                     int minScaleLevel = -1, int maxScaleLevel = INT_MAX) {
        const int n1 = (int)F1.mvpMapPoints.size(), n2 = (int)F2.mvKeysUn.size();
        vpMapPointMatches2.assign(F2.mvpMapPoints.size(), static_cast<MapPointT*>(0));
        std::vector<unsigned char> valid(n1 ? n1 : 1);
        for (int i = 0; i < n1; i++) valid[i] = F1.mvpMapPoints[i] && !F1.mvpMapPoints[i]->isBad();
        const std::vector<unsigned char> d1 = packDescriptors(F1), d2 = packDescriptors(F2);
        const OrbFrameBounds b = {FrameT::mnMinX, FrameT::mnMinY, FrameT::mnMaxX, FrameT::mnMaxY};
        std::vector<int> vnMatches21(n2 ? n2 : 1, -1);
        int nmatches = 0;
        if (orbhip_window_search(handle, (const OrbKeypoint*)F1.mvKeysUn.data(), d1.data(), n1, valid.data(),
                                 (const OrbKeypoint*)F2.mvKeysUn.data(), d2.data(), n2, b, windowSize, minScaleLevel,
                                 maxScaleLevel, mfNNratio, mbCheckOrientation, vnMatches21.data(), &nmatches) != ORBHIP_OK)
            throw std::runtime_error(orbhip_last_error());
        for (int i2 = 0; i2 < n2; i2++)
            if (vnMatches21[i2] >= 0) vpMapPointMatches2[i2] = F1.mvpMapPoints[vnMatches21[i2]];
        return nmatches;
    }

    // int SearchByProjection(Frame &F, const std::vector<MapPoint*> &vpMapPoints, const float th = 3)
    // (include/ORBmatcher.h:54, src/ORBmatcher.cc:52-128): the lines that read the tracking fields of a map
    // point stay as in the reference and fill one probe each; the search and assignment loops run on the
    // device.  FrameT additionally needs mvScaleFactors; MapPointT needs mbTrackInView, mnTrackScaleLevel,
    // mTrackViewCos, mTrackProjX, mTrackProjY and GetDescriptor().
    static float RadiusByViewingCos(const float& viewCos) {
        if (viewCos > 0.998)
            return 2.5;
        else
            return 4.0;
    }
    template <class FrameT, class MapPointT>
    int SearchByProjection(FrameT& F, const std::vector<MapPointT*>& vpMapPoints, const float th = 3) {
        const int n = (int)F.mvKeysUn.size(), nmp = (int)vpMapPoints.size();
        std::vector<OrbProbe> probes(nmp ? nmp : 1);
        std::vector<unsigned char> pdesc((size_t)(nmp ? nmp : 1) * 32);
        const bool bFactor = th != 1.0;
        for (int iMP = 0; iMP < nmp; iMP++) {
            OrbProbe& q = probes[iMP];
            q.u = q.v = q.r = q.angle = 0.f;
            q.min_level = q.max_level = q.valid = q.reserved = 0;
            MapPointT* pMP = vpMapPoints[iMP];
            if (!pMP->mbTrackInView) continue;
            if (pMP->isBad()) continue;
            const int& nPredictedLevel = pMP->mnTrackScaleLevel;
            float r = RadiusByViewingCos(pMP->mTrackViewCos);
            if (bFactor) r *= th;
            q.u = pMP->mTrackProjX;
            q.v = pMP->mTrackProjY;
            q.r = r * F.mvScaleFactors[nPredictedLevel];
            q.min_level = nPredictedLevel - 1;
            q.max_level = nPredictedLevel;
            q.valid = 1;
            cv::Mat MPdescriptor = pMP->GetDescriptor();
            const unsigned char* d = MPdescriptor.template ptr<unsigned char>(0);
            for (int b = 0; b < 32; b++) pdesc[(size_t)iMP * 32 + b] = d[b];
        }
        std::vector<unsigned char> taken(n ? n : 1);
        for (int i = 0; i < n; i++) taken[i] = F.mvpMapPoints[i] != 0;
        const std::vector<unsigned char> d2 = packDescriptors(F);
        const OrbFrameBounds b = {FrameT::mnMinX, FrameT::mnMinY, FrameT::mnMaxX, FrameT::mnMaxY};
        std::vector<int> owner(n ? n : 1, -1);
        int nmatches = 0;
        if (orbhip_search_by_projection(handle, probes.data(), pdesc.data(), nmp, (const OrbKeypoint*)F.mvKeysUn.data(),
                                        d2.data(), n, b, taken.data(), ORBHIP_PROJ_LOCAL_MAP, mfNNratio, TH_HIGH, 0,
                                        owner.data(), &nmatches) != ORBHIP_OK)
            throw std::runtime_error(orbhip_last_error());
        for (int i = 0; i < n; i++)
            if (owner[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[owner[i]];
        return nmatches;
    }
};
}  // namespace ORB_SLAM
#endif
