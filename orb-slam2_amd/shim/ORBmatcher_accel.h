// The accelerated primitive of the reference's ORBmatcher, with its signature unchanged:
//   static int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b)
// (include/ORBmatcher.h:46, src/ORBmatcher.cc:1805-1821).  In the reference tree the body of that
// one function is replaced by the single forwarding line below; the search methods get the forwarding bodies that
// follow.  Batched dense matching goes through orbhip_knn2*.
#ifndef ORBMATCHER_ACCEL_H
#define ORBMATCHER_ACCEL_H
#ifdef ORBHIP_SHIM_CV_STUB
#include "cv_stub.h"
#else
#include <opencv2/core/core.hpp>
#endif
#include <algorithm>
#include <climits>
#include <cmath>
#include <set>
#include <stdexcept>
#include <utility>
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

    // ---- shared by the projection overloads below --------------------------------------------------
    static void ClearProbe(OrbProbe& q) {
        q.u = q.v = q.r = q.angle = 0.f;
        q.min_level = q.max_level = q.valid = q.reserved = 0;
    }
    static void CopyDescriptor(const cv::Mat& d, unsigned char* dst) {
        const unsigned char* p = d.ptr<unsigned char>(0);
        for (int b = 0; b < 32; b++) dst[b] = p[b];
    }
    static std::vector<unsigned char> packRows(const cv::Mat& D) {
        std::vector<unsigned char> d((size_t)(D.rows ? D.rows : 1) * 32);
        for (int i = 0; i < D.rows; i++) CopyDescriptor(D.row(i), &d[(size_t)i * 32]);
        return d;
    }
    // probes -> device search -> owner[i2] = probe index assigned to feature i2 (or -1)
    int RunProbes(const std::vector<OrbProbe>& probes, const std::vector<unsigned char>& pdesc, int nprobes,
                  const std::vector<cv::KeyPoint>& keys, const std::vector<unsigned char>& desc, const OrbFrameBounds& b,
                  const std::vector<unsigned char>& taken, int rule, int thDist, bool checkOri, std::vector<int>& owner) {
        const int n = (int)keys.size();
        owner.assign(n ? n : 1, -1);
        int nmatches = 0;
        if (orbhip_search_by_projection(handle, probes.data(), pdesc.data(), nprobes, (const OrbKeypoint*)keys.data(), desc.data(),
                                        n, b, taken.data(), rule, mfNNratio, thDist, checkOri, owner.data(), &nmatches) != ORBHIP_OK)
            throw std::runtime_error(orbhip_last_error());
        return nmatches;
    }

    // int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th)
    // (include/ORBmatcher.h:58, src/ORBmatcher.cc:297-418).  The pose decomposition and the per-point projection,
    // depth, distance and viewing-angle tests stay the reference's lines; every point that passes them becomes a
    // probe.  FrameT only supplies the image bounds every KeyFrame copies from it (src/KeyFrame.cc:44).
    template <class FrameT, class KeyFrameT, class MapPointT>
    int SearchByProjection(KeyFrameT* pKF, cv::Mat Scw, const std::vector<MapPointT*>& vpPoints,
                           std::vector<MapPointT*>& vpMatched, int th) {
        const float fx = pKF->fx, fy = pKF->fy, cx = pKF->cx, cy = pKF->cy;
        const int nMaxLevel = pKF->GetScaleLevels() - 1;
        std::vector<float> vfScaleFactors = pKF->GetScaleFactors();
        cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
        const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
        cv::Mat Rcw = sRcw / scw;
        cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
        cv::Mat Ow = -Rcw.t() * tcw;
        std::set<MapPointT*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
        spAlreadyFound.erase(static_cast<MapPointT*>(0));

        const int nmp = (int)vpPoints.size();
        std::vector<OrbProbe> probes(nmp ? nmp : 1);
        std::vector<unsigned char> pdesc((size_t)(nmp ? nmp : 1) * 32);
        for (int iMP = 0; iMP < nmp; iMP++) {
            OrbProbe& q = probes[iMP];
            ClearProbe(q);
            MapPointT* pMP = vpPoints[iMP];
            if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
            cv::Mat p3Dw = pMP->GetWorldPos();
            cv::Mat p3Dc = Rcw * p3Dw + tcw;
            if (p3Dc.template at<float>(2) < 0.0) continue;
            const float invz = 1 / p3Dc.template at<float>(2);
            const float x = p3Dc.template at<float>(0) * invz;
            const float y = p3Dc.template at<float>(1) * invz;
            const float u = fx * x + cx;
            const float v = fy * y + cy;
            if (!pKF->IsInImage(u, v)) continue;
            const float maxDistance = pMP->GetMaxDistanceInvariance();
            const float minDistance = pMP->GetMinDistanceInvariance();
            cv::Mat PO = p3Dw - Ow;
            const float dist = cv::norm(PO);
            if (dist < minDistance || dist > maxDistance) continue;
            cv::Mat Pn = pMP->GetNormal();
            if (PO.dot(Pn) < 0.5 * dist) continue;
            const float ratio = dist / minDistance;
            std::vector<float>::iterator it = std::lower_bound(vfScaleFactors.begin(), vfScaleFactors.end(), ratio);
            const int nPredictedLevel = std::min(static_cast<int>(it - vfScaleFactors.begin()), nMaxLevel);
            q.u = u;
            q.v = v;
            q.r = th * pKF->GetScaleFactor(nPredictedLevel);
            q.min_level = nPredictedLevel - 1;  // the kpLevel test of :392-395
            q.max_level = nPredictedLevel;
            q.valid = 1;
            CopyDescriptor(pMP->GetDescriptor(), &pdesc[(size_t)iMP * 32]);
        }
        const std::vector<cv::KeyPoint> keys = pKF->GetKeyPointsUn();
        const std::vector<unsigned char> desc = packRows(pKF->GetDescriptors());
        const int n = (int)keys.size();
        std::vector<unsigned char> taken(n ? n : 1);
        for (int i = 0; i < n; i++) taken[i] = vpMatched[i] != 0;
        const OrbFrameBounds b = {FrameT::mnMinX, FrameT::mnMinY, FrameT::mnMaxX, FrameT::mnMaxY};
        std::vector<int> owner;
        const int nmatches = RunProbes(probes, pdesc, nmp, keys, desc, b, taken, ORBHIP_PROJ_BEST, TH_LOW, false, owner);
        for (int i = 0; i < n; i++)
            if (owner[i] >= 0) vpMatched[i] = vpPoints[owner[i]];
        return nmatches;
    }

    // int SearchByProjection(Frame &F1, Frame &F2, int windowSize, vector<MapPoint*> &vpMapPointMatches2)
    // (include/ORBmatcher.h:66, src/ORBmatcher.cc:530-605)
    template <class FrameT, class MapPointT>
    int SearchByProjection(FrameT& F1, FrameT& F2, int windowSize, std::vector<MapPointT*>& vpMapPointMatches2) {
        vpMapPointMatches2 = F2.mvpMapPoints;
        std::set<MapPointT*> spMapPointsAlreadyFound(vpMapPointMatches2.begin(), vpMapPointMatches2.end());
        const cv::Mat Rc2w = F2.mTcw.rowRange(0, 3).colRange(0, 3);
        const cv::Mat tc2w = F2.mTcw.rowRange(0, 3).col(3);
        const int n1 = (int)F1.mvpMapPoints.size();
        std::vector<OrbProbe> probes(n1 ? n1 : 1);
        for (int i1 = 0; i1 < n1; i1++) {
            OrbProbe& q = probes[i1];
            ClearProbe(q);
            MapPointT* pMP1 = F1.mvpMapPoints[i1];
            if (!pMP1) continue;
            if (pMP1->isBad() || spMapPointsAlreadyFound.count(pMP1)) continue;
            cv::KeyPoint kp1 = F1.mvKeysUn[i1];
            int level1 = kp1.octave;
            cv::Mat x3Dw = pMP1->GetWorldPos();
            cv::Mat x3Dc2 = Rc2w * x3Dw + tc2w;
            const float xc2 = x3Dc2.template at<float>(0);
            const float yc2 = x3Dc2.template at<float>(1);
            const float invzc2 = 1.0 / x3Dc2.template at<float>(2);
            q.u = FrameT::fx * xc2 * invzc2 + FrameT::cx;
            q.v = FrameT::fy * yc2 * invzc2 + FrameT::cy;
            q.r = windowSize;
            q.min_level = q.max_level = level1;
            q.valid = 1;
        }
        const std::vector<unsigned char> pdesc = packDescriptors(F1), d2 = packDescriptors(F2);
        const int n2 = (int)F2.mvKeysUn.size();
        std::vector<unsigned char> taken(n2 ? n2 : 1);
        for (int i = 0; i < n2; i++) taken[i] = vpMapPointMatches2[i] != 0;
        const OrbFrameBounds b = {FrameT::mnMinX, FrameT::mnMinY, FrameT::mnMaxX, FrameT::mnMaxY};
        std::vector<int> owner;
        const int nmatches = RunProbes(probes, pdesc, n1, F2.mvKeysUn, d2, b, taken, ORBHIP_PROJ_RATIO, TH_HIGH, false, owner);
        for (int i = 0; i < n2; i++)
            if (owner[i] >= 0) vpMapPointMatches2[i] = F1.mvpMapPoints[owner[i]];
        return nmatches;
    }

    // int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, float th)
    // (include/ORBmatcher.h:81, src/ORBmatcher.cc:1518-1631)
    template <class FrameT>
    int SearchByProjection(FrameT& CurrentFrame, const FrameT& LastFrame, float th) {
        const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
        const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
        const int nl = (int)LastFrame.mvpMapPoints.size();
        std::vector<OrbProbe> probes(nl ? nl : 1);
        for (int i = 0; i < nl; i++) {
            OrbProbe& q = probes[i];
            ClearProbe(q);
            if (!LastFrame.mvpMapPoints[i]) continue;
            if (LastFrame.mvbOutlier[i]) continue;
            cv::Mat x3Dw = LastFrame.mvpMapPoints[i]->GetWorldPos();
            cv::Mat x3Dc = Rcw * x3Dw + tcw;
            const float xc = x3Dc.template at<float>(0);
            const float yc = x3Dc.template at<float>(1);
            const float invzc = 1.0 / x3Dc.template at<float>(2);
            float u = FrameT::fx * xc * invzc + FrameT::cx;
            float v = FrameT::fy * yc * invzc + FrameT::cy;
            if (u < FrameT::mnMinX || u > FrameT::mnMaxX) continue;
            if (v < FrameT::mnMinY || v > FrameT::mnMaxY) continue;
            int nPredictedOctave = LastFrame.mvKeys[i].octave;
            q.u = u;
            q.v = v;
            q.r = th * CurrentFrame.mvScaleFactors[nPredictedOctave];
            q.min_level = nPredictedOctave - 1;
            q.max_level = nPredictedOctave + 1;
            q.angle = LastFrame.mvKeysUn[i].angle;
            q.valid = 1;
        }
        const std::vector<unsigned char> pdesc = packDescriptors(LastFrame), d2 = packDescriptors(CurrentFrame);
        const int n = (int)CurrentFrame.mvKeysUn.size();
        std::vector<unsigned char> taken(n ? n : 1);
        for (int i = 0; i < n; i++) taken[i] = CurrentFrame.mvpMapPoints[i] != 0;
        const OrbFrameBounds b = {FrameT::mnMinX, FrameT::mnMinY, FrameT::mnMaxX, FrameT::mnMaxY};
        std::vector<int> owner;
        const int nmatches =
            RunProbes(probes, pdesc, nl, CurrentFrame.mvKeysUn, d2, b, taken, ORBHIP_PROJ_BEST, TH_HIGH, mbCheckOrientation, owner);
        for (int i = 0; i < n; i++)
            if (owner[i] >= 0) CurrentFrame.mvpMapPoints[i] = LastFrame.mvpMapPoints[owner[i]];
        return nmatches;
    }

    // int SearchByProjection(Frame &CurrentFrame, KeyFrame* pKF, const set<MapPoint*> &sAlreadyFound, float th, int ORBdist)
    // (include/ORBmatcher.h:85, src/ORBmatcher.cc:1633-1757)
    template <class FrameT, class KeyFrameT, class MapPointT>
    int SearchByProjection(FrameT& CurrentFrame, KeyFrameT* pKF, const std::set<MapPointT*>& sAlreadyFound, float th, int ORBdist) {
        const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
        const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
        const cv::Mat Ow = -Rcw.t() * tcw;
        std::vector<MapPointT*> vpMPs = pKF->GetMapPointMatches();
        const int nk = (int)vpMPs.size();
        std::vector<OrbProbe> probes(nk ? nk : 1);
        std::vector<unsigned char> pdesc((size_t)(nk ? nk : 1) * 32);
        for (int i = 0; i < nk; i++) {
            OrbProbe& q = probes[i];
            ClearProbe(q);
            MapPointT* pMP = vpMPs[i];
            if (!pMP) continue;
            if (pMP->isBad() || sAlreadyFound.count(pMP)) continue;
            cv::Mat x3Dw = pMP->GetWorldPos();
            cv::Mat x3Dc = Rcw * x3Dw + tcw;
            const float xc = x3Dc.template at<float>(0);
            const float yc = x3Dc.template at<float>(1);
            const float invzc = 1.0 / x3Dc.template at<float>(2);
            float u = FrameT::fx * xc * invzc + FrameT::cx;
            float v = FrameT::fy * yc * invzc + FrameT::cy;
            if (u < FrameT::mnMinX || u > FrameT::mnMaxX) continue;
            if (v < FrameT::mnMinY || v > FrameT::mnMaxY) continue;
            float minDistance = pMP->GetMinDistanceInvariance();
            cv::Mat PO = x3Dw - Ow;
            float dist3D = cv::norm(PO);
            float ratio = dist3D / minDistance;
            std::vector<float>::iterator it =
                std::lower_bound(CurrentFrame.mvScaleFactors.begin(), CurrentFrame.mvScaleFactors.end(), ratio);
            const int nPredictedLevel =
                std::min(static_cast<int>(it - CurrentFrame.mvScaleFactors.begin()), CurrentFrame.mnScaleLevels - 1);
            q.u = u;
            q.v = v;
            q.r = th * CurrentFrame.mvScaleFactors[nPredictedLevel];
            q.min_level = nPredictedLevel - 1;
            q.max_level = nPredictedLevel + 1;
            q.angle = pKF->GetKeyPointUn(i).angle;
            q.valid = 1;
            CopyDescriptor(pMP->GetDescriptor(), &pdesc[(size_t)i * 32]);
        }
        const std::vector<unsigned char> d2 = packDescriptors(CurrentFrame);
        const int n = (int)CurrentFrame.mvKeysUn.size();
        std::vector<unsigned char> taken(n ? n : 1);
        for (int i = 0; i < n; i++) taken[i] = CurrentFrame.mvpMapPoints[i] != 0;
        const OrbFrameBounds b = {FrameT::mnMinX, FrameT::mnMinY, FrameT::mnMaxX, FrameT::mnMaxY};
        std::vector<int> owner;
        const int nmatches =
            RunProbes(probes, pdesc, nk, CurrentFrame.mvKeysUn, d2, b, taken, ORBHIP_PROJ_BEST, ORBdist, mbCheckOrientation, owner);
        for (int i = 0; i < n; i++)
            if (owner[i] >= 0) CurrentFrame.mvpMapPoints[i] = vpMPs[owner[i]];
        return nmatches;
    }

    // ---- Fuse and SearchBySim3 ------------------------------------------------------------------------
    // bestIdx / bestDist of every probe among the features of one keyframe ("match to the most similar keypoint
    // in the radius", src/ORBmatcher.cc:1096-1124): -1 / INT_MAX where the probe is not valid or nothing passes.
    template <class FrameT, class KeyFrameT>
    void BestInRadius(KeyFrameT* pKF, const std::vector<OrbProbe>& probes, const std::vector<unsigned char>& pdesc, int nprobes,
                      std::vector<int>& bestIdx, std::vector<int>& bestDist) {
        const std::vector<cv::KeyPoint> keys = pKF->GetKeyPointsUn();
        const std::vector<unsigned char> desc = packRows(pKF->GetDescriptors());
        const OrbFrameBounds b = {FrameT::mnMinX, FrameT::mnMinY, FrameT::mnMaxX, FrameT::mnMaxY};
        bestIdx.assign(nprobes ? nprobes : 1, -1);
        bestDist.assign(nprobes ? nprobes : 1, INT_MAX);
        if (orbhip_best_in_radius(handle, probes.data(), pdesc.data(), nprobes, (const OrbKeypoint*)keys.data(), desc.data(),
                                  (int)keys.size(), b, bestIdx.data(), bestDist.data()) != ORBHIP_OK)
            throw std::runtime_error(orbhip_last_error());
    }

    // The lines of Fuse from "Depth must be positive" to the radius (src/ORBmatcher.cc:1057-1094, :1185-1223) for one
    // map point; false where the reference takes a `continue`.
    template <class KeyFrameT, class MapPointT>
    static bool FuseProbe(KeyFrameT* pKF, MapPointT* pMP, const cv::Mat& Rcw, const cv::Mat& tcw, const cv::Mat& Ow,
                          const std::vector<float>& vfScaleFactors, float th, OrbProbe& q) {
        ClearProbe(q);
        const float &fx = pKF->fx, &fy = pKF->fy, &cx = pKF->cx, &cy = pKF->cy;
        const int nMaxLevel = pKF->GetScaleLevels() - 1;
        cv::Mat p3Dw = pMP->GetWorldPos();
        cv::Mat p3Dc = Rcw * p3Dw + tcw;
        if (p3Dc.template at<float>(2) < 0.0f) return false;
        const float invz = 1 / p3Dc.template at<float>(2);
        const float x = p3Dc.template at<float>(0) * invz;
        const float y = p3Dc.template at<float>(1) * invz;
        const float u = fx * x + cx;
        const float v = fy * y + cy;
        if (!pKF->IsInImage(u, v)) return false;
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        cv::Mat PO = p3Dw - Ow;
        const float dist3D = cv::norm(PO);
        if (dist3D < minDistance || dist3D > maxDistance) return false;
        cv::Mat Pn = pMP->GetNormal();
        if (PO.dot(Pn) < 0.5 * dist3D) return false;
        const float ratio = dist3D / minDistance;
        std::vector<float>::const_iterator it = std::lower_bound(vfScaleFactors.begin(), vfScaleFactors.end(), ratio);
        const int nPredictedLevel = std::min(static_cast<int>(it - vfScaleFactors.begin()), nMaxLevel);
        q.u = u;
        q.v = v;
        q.r = th * vfScaleFactors[nPredictedLevel];
        q.min_level = nPredictedLevel - 1;  // the kpLevel test of :1112-1113 / :1241-1242
        q.max_level = nPredictedLevel;
        q.valid = 1;
        return true;
    }

    // Both forms of Fuse.  Pass 1 projects every map point and searches all of them in one device call; pass 2 is
    // the reference's loop with its pointer tests, Replace / AddObservation / AddMapPoint in the reference's order
    // (they change what later iterations see: GetMapPoint(bestIdx), isBad(), IsInKeyFrame).  A map point that
    // occurs a second time in the list is searched again at that moment, because a Replace in between may have
    // given it another descriptor (MapPoint::Replace ends in ComputeDistinctiveDescriptors, src/MapPoint.cc:160).
    template <class FrameT, class KeyFrameT, class MapPointT, class SkipT, class FuseT>
    int FuseLoop(KeyFrameT* pKF, const std::vector<MapPointT*>& vpMapPoints, const cv::Mat& Rcw, const cv::Mat& tcw,
                 const cv::Mat& Ow, float th, SkipT skip, FuseT fuse) {
        const std::vector<float> vfScaleFactors = pKF->GetScaleFactors();
        const int nmp = (int)vpMapPoints.size();
        std::vector<OrbProbe> probes(nmp ? nmp : 1);
        std::vector<unsigned char> pdesc((size_t)(nmp ? nmp : 1) * 32);
        for (int i = 0; i < nmp; i++) {
            ClearProbe(probes[i]);
            MapPointT* pMP = vpMapPoints[i];
            if (!pMP) continue;
            if (FuseProbe(pKF, pMP, Rcw, tcw, Ow, vfScaleFactors, th, probes[i])) CopyDescriptor(pMP->GetDescriptor(), &pdesc[(size_t)i * 32]);
        }
        std::vector<int> vBestIdx, vBestDist;
        BestInRadius<FrameT>(pKF, probes, pdesc, nmp, vBestIdx, vBestDist);
        std::set<MapPointT*> seen;
        int nFused = 0;
        for (int i = 0; i < nmp; i++) {
            MapPointT* pMP = vpMapPoints[i];
            if (!pMP) continue;
            if (skip(pMP)) continue;
            int bestDist = vBestDist[i];
            int bestIdx = vBestIdx[i];
            if (!seen.insert(pMP).second) {
                std::vector<OrbProbe> one(1);
                std::vector<unsigned char> d(32);
                std::vector<int> bi, bd;
                if (FuseProbe(pKF, pMP, Rcw, tcw, Ow, vfScaleFactors, th, one[0])) CopyDescriptor(pMP->GetDescriptor(), &d[0]);
                BestInRadius<FrameT>(pKF, one, d, 1, bi, bd);
                bestIdx = bi[0];
                bestDist = bd[0];
            }
            // If there is already a MapPoint replace otherwise add new measurement
            if (bestDist <= TH_LOW) {
                fuse(pMP, bestIdx);
                nFused++;
            }
        }
        return nFused;
    }

    // int Fuse(KeyFrame* pKF, std::vector<MapPoint *> &vpMapPoints, float th=2.5)
    // (include/ORBmatcher.h:87, src/ORBmatcher.cc:1027-1145)
    template <class FrameT, class KeyFrameT, class MapPointT>
    int Fuse(KeyFrameT* pKF, std::vector<MapPointT*>& vpMapPoints, float th = 2.5) {
        cv::Mat Rcw = pKF->GetRotation();
        cv::Mat tcw = pKF->GetTranslation();
        cv::Mat Ow = pKF->GetCameraCenter();
        return FuseLoop<FrameT>(
            pKF, vpMapPoints, Rcw, tcw, Ow, th, [&](MapPointT* pMP) { return pMP->isBad() || pMP->IsInKeyFrame(pKF); },
            [&](MapPointT* pMP, int bestIdx) {
                MapPointT* pMPinKF = pKF->GetMapPoint(bestIdx);
                if (pMPinKF) {
                    if (!pMPinKF->isBad()) pMP->Replace(pMPinKF);
                } else {
                    pMP->AddObservation(pKF, bestIdx);
                    pKF->AddMapPoint(pMP, bestIdx);
                }
            });
    }

    // int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*> &vpPoints, float th=2.5)
    // (include/ORBmatcher.h:90, src/ORBmatcher.cc:1147-1276)
    template <class FrameT, class KeyFrameT, class MapPointT>
    int Fuse(KeyFrameT* pKF, cv::Mat Scw, const std::vector<MapPointT*>& vpPoints, float th = 2.5) {
        cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
        const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
        cv::Mat Rcw = sRcw / scw;
        cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
        cv::Mat Ow = -Rcw.t() * tcw;
        std::set<MapPointT*> spAlreadyFound = pKF->GetMapPoints();
        return FuseLoop<FrameT>(
            pKF, vpPoints, Rcw, tcw, Ow, th, [&](MapPointT* pMP) { return pMP->isBad() || spAlreadyFound.count(pMP); },
            [&](MapPointT* pMP, int bestIdx) {
                MapPointT* pMPinKF = pKF->GetMapPoint(bestIdx);
                if (pMPinKF) {
                    if (!pMPinKF->isBad()) pMPinKF->Replace(pMP);
                } else {
                    pMP->AddObservation(pKF, bestIdx);
                    pKF->AddMapPoint(pMP, bestIdx);
                }
            });
    }

    // int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12,
    //                  const cv::Mat &R12, const cv::Mat &t12, float th)
    // (include/ORBmatcher.h:84, src/ORBmatcher.cc:1278-1516): the reference's lines up to each GetFeaturesInArea
    // build the probes of both directions, two device searches, then the reference's agreement check.
    template <class FrameT, class KeyFrameT, class MapPointT>
    int SearchBySim3(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12, const float& s12,
                     const cv::Mat& R12, const cv::Mat& t12, float th) {
        const float fx = pKF1->fx;
        const float fy = pKF1->fy;
        const float cx = pKF1->cx;
        const float cy = pKF1->cy;
        cv::Mat R1w = pKF1->GetRotation();
        cv::Mat t1w = pKF1->GetTranslation();
        cv::Mat R2w = pKF2->GetRotation();
        cv::Mat t2w = pKF2->GetTranslation();
        cv::Mat sR12 = s12 * R12;
        cv::Mat sR21 = (1.0 / s12) * R12.t();
        cv::Mat t21 = -sR21 * t12;
        const int nMaxLevel1 = pKF1->GetScaleLevels() - 1;
        std::vector<float> vfScaleFactors1 = pKF1->GetScaleFactors();
        std::vector<MapPointT*> vpMapPoints1 = pKF1->GetMapPointMatches();
        const int N1 = vpMapPoints1.size();
        const int nMaxLevel2 = pKF2->GetScaleLevels() - 1;
        std::vector<float> vfScaleFactors2 = pKF2->GetScaleFactors();
        std::vector<MapPointT*> vpMapPoints2 = pKF2->GetMapPointMatches();
        const int N2 = vpMapPoints2.size();
        std::vector<bool> vbAlreadyMatched1(N1, false);
        std::vector<bool> vbAlreadyMatched2(N2, false);
        for (int i = 0; i < N1; i++) {
            MapPointT* pMP = vpMatches12[i];
            if (pMP) {
                vbAlreadyMatched1[i] = true;
                int idx2 = pMP->GetIndexInKeyFrame(pKF2);
                if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
            }
        }
        // one direction: the map points of `from` (already in the camera of `from`: Rfw, tfw), moved into the camera
        // of `to` by sR * p + t and searched among the features of `to`
        struct Direction {
            static void Probes(const std::vector<MapPointT*>& vpMapPoints, const std::vector<bool>& vbAlreadyMatched,
                               const cv::Mat& Rfw, const cv::Mat& tfw, const cv::Mat& sR, const cv::Mat& t, KeyFrameT* pKFto,
                               const std::vector<float>& vfScaleFactorsTo, int nMaxLevelTo, float fx, float fy, float cx, float cy,
                               float th, std::vector<OrbProbe>& probes, std::vector<unsigned char>& pdesc) {
                const int N = (int)vpMapPoints.size();
                probes.resize(N ? N : 1);
                pdesc.assign((size_t)(N ? N : 1) * 32, 0);
                for (int i = 0; i < N; i++) {
                    OrbProbe& q = probes[i];
                    ClearProbe(q);
                    MapPointT* pMP = vpMapPoints[i];
                    if (!pMP || vbAlreadyMatched[i]) continue;
                    if (pMP->isBad()) continue;
                    cv::Mat p3Dw = pMP->GetWorldPos();
                    cv::Mat p3Dcf = Rfw * p3Dw + tfw;
                    cv::Mat p3Dct = sR * p3Dcf + t;
                    if (p3Dct.template at<float>(2) < 0.0) continue;
                    float invz = 1.0 / p3Dct.template at<float>(2);
                    float x = p3Dct.template at<float>(0) * invz;
                    float y = p3Dct.template at<float>(1) * invz;
                    float u = fx * x + cx;
                    float v = fy * y + cy;
                    if (!pKFto->IsInImage(u, v)) continue;
                    float maxDistance = pMP->GetMaxDistanceInvariance();
                    float minDistance = pMP->GetMinDistanceInvariance();
                    float dist3D = cv::norm(p3Dct);
                    if (dist3D < minDistance || dist3D > maxDistance) continue;
                    float ratio = dist3D / minDistance;
                    std::vector<float>::const_iterator it = std::lower_bound(vfScaleFactorsTo.begin(), vfScaleFactorsTo.end(), ratio);
                    const int nPredictedLevel = std::min(static_cast<int>(it - vfScaleFactorsTo.begin()), nMaxLevelTo);
                    q.u = u;
                    q.v = v;
                    q.r = th * vfScaleFactorsTo[nPredictedLevel];
                    q.min_level = nPredictedLevel - 1;  // the kp.octave test of :1392-1393 / :1477-1478
                    q.max_level = nPredictedLevel;
                    q.valid = 1;
                    CopyDescriptor(pMP->GetDescriptor(), &pdesc[(size_t)i * 32]);
                }
            }
        };
        std::vector<OrbProbe> probes1, probes2;
        std::vector<unsigned char> pdesc1, pdesc2;
        Direction::Probes(vpMapPoints1, vbAlreadyMatched1, R1w, t1w, sR21, t21, pKF2, vfScaleFactors2, nMaxLevel2, fx, fy, cx, cy, th,
                          probes1, pdesc1);
        Direction::Probes(vpMapPoints2, vbAlreadyMatched2, R2w, t2w, sR12, t12, pKF1, vfScaleFactors1, nMaxLevel1, fx, fy, cx, cy, th,
                          probes2, pdesc2);
        std::vector<int> bestIdx1, bestDist1, bestIdx2, bestDist2;
        BestInRadius<FrameT>(pKF2, probes1, pdesc1, N1, bestIdx1, bestDist1);
        BestInRadius<FrameT>(pKF1, probes2, pdesc2, N2, bestIdx2, bestDist2);
        std::vector<int> vnMatch1(N1, -1);
        std::vector<int> vnMatch2(N2, -1);
        for (int i1 = 0; i1 < N1; i1++)
            if (bestDist1[i1] <= TH_HIGH) vnMatch1[i1] = bestIdx1[i1];
        for (int i2 = 0; i2 < N2; i2++)
            if (bestDist2[i2] <= TH_HIGH) vnMatch2[i2] = bestIdx2[i2];
        // Check agreement
        int nFound = 0;
        for (int i1 = 0; i1 < N1; i1++) {
            int idx2 = vnMatch1[i1];
            if (idx2 >= 0) {
                int idx1 = vnMatch2[idx2];
                if (idx1 == i1) {
                    vpMatches12[i1] = vpMapPoints2[idx2];
                    nFound++;
                }
            }
        }
        return nFound;
    }

    // ---- SearchByBoW --------------------------------------------------------------------------------
    // A DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned int>>) flattened in map order.
    struct FlatFeatureVector {
        std::vector<int> nodes, start, feat;
        template <class FeatVecT>
        explicit FlatFeatureVector(const FeatVecT& fv) {
            start.push_back(0);
            for (typename FeatVecT::const_iterator it = fv.begin(); it != fv.end(); ++it) {
                nodes.push_back((int)it->first);
                for (size_t k = 0; k < it->second.size(); k++) feat.push_back((int)it->second[k]);
                start.push_back((int)feat.size());
            }
            if (nodes.empty()) nodes.push_back(0);
            if (feat.empty()) feat.push_back(0);
        }
        int count() const { return (int)start.size() - 1; }
    };

    // int SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches)
    // (include/ORBmatcher.h:62, src/ORBmatcher.cc:158-295)
    template <class KeyFrameT, class FrameT, class MapPointT>
    int SearchByBoW(KeyFrameT* pKF, FrameT& F, std::vector<MapPointT*>& vpMapPointMatches) {
        std::vector<MapPointT*> vpMapPointsKF = pKF->GetMapPointMatches();
        vpMapPointMatches = std::vector<MapPointT*>(F.mvpMapPoints.size(), static_cast<MapPointT*>(0));
        const FlatFeatureVector fvKF(pKF->GetFeatureVector()), fvF(F.mFeatVec);
        const std::vector<cv::KeyPoint> keysKF = pKF->GetKeyPointsUn();
        const std::vector<unsigned char> dKF = packRows(pKF->GetDescriptors()), dF = packDescriptors(F);
        const int nKF = (int)keysKF.size(), nF = (int)F.mvKeys.size();
        // the .at() of :190-196: a feature index past the keyframe's map points is skipped like a NULL one
        std::vector<unsigned char> validKF(nKF ? nKF : 1);
        for (int i = 0; i < nKF; i++)
            validKF[i] = (size_t)i < vpMapPointsKF.size() && vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();
        std::vector<int> m12(nKF ? nKF : 1, -1), m21(nF ? nF : 1, -1);
        int nmatches = 0;
        if (orbhip_search_by_bow(handle, (const OrbKeypoint*)keysKF.data(), dKF.data(), nKF, validKF.data(), fvKF.nodes.data(),
                                 fvKF.start.data(), fvKF.feat.data(), fvKF.count(), (const OrbKeypoint*)F.mvKeys.data(), dF.data(), nF,
                                 0, fvF.nodes.data(), fvF.start.data(), fvF.feat.data(), fvF.count(), mfNNratio, mbCheckOrientation,
                                 1, m12.data(), m21.data(), &nmatches) != ORBHIP_OK)
            throw std::runtime_error(orbhip_last_error());
        for (int iF = 0; iF < nF; iF++)
            if (m21[iF] >= 0) vpMapPointMatches[iF] = vpMapPointsKF[m21[iF]];
        return nmatches;
    }

    // int SearchByBoW(KeyFrame *pKF1, KeyFrame* pKF2, vector<MapPoint*> &vpMatches12)
    // (include/ORBmatcher.h:63, src/ORBmatcher.cc:726-861)
    template <class KeyFrameT, class MapPointT>
    int SearchByBoW(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12) {
        std::vector<cv::KeyPoint> vKeysUn1 = pKF1->GetKeyPointsUn(), vKeysUn2 = pKF2->GetKeyPointsUn();
        const FlatFeatureVector fv1(pKF1->GetFeatureVector()), fv2(pKF2->GetFeatureVector());
        std::vector<MapPointT*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
        const std::vector<unsigned char> d1 = packRows(pKF1->GetDescriptors()), d2 = packRows(pKF2->GetDescriptors());
        vpMatches12 = std::vector<MapPointT*>(vpMapPoints1.size(), static_cast<MapPointT*>(0));
        const int n1 = (int)vKeysUn1.size(), n2 = (int)vKeysUn2.size();
        std::vector<unsigned char> valid1(n1 ? n1 : 1), valid2(n2 ? n2 : 1);
        for (int i = 0; i < n1; i++) valid1[i] = vpMapPoints1[i] && !vpMapPoints1[i]->isBad();
        for (int i = 0; i < n2; i++) valid2[i] = vpMapPoints2[i] && !vpMapPoints2[i]->isBad();
        std::vector<int> m12(n1 ? n1 : 1, -1), m21(n2 ? n2 : 1, -1);
        int nmatches = 0;
        if (orbhip_search_by_bow(handle, (const OrbKeypoint*)vKeysUn1.data(), d1.data(), n1, valid1.data(), fv1.nodes.data(),
                                 fv1.start.data(), fv1.feat.data(), fv1.count(), (const OrbKeypoint*)vKeysUn2.data(), d2.data(), n2,
                                 valid2.data(), fv2.nodes.data(), fv2.start.data(), fv2.feat.data(), fv2.count(), mfNNratio,
                                 mbCheckOrientation, 0, m12.data(), m21.data(), &nmatches) != ORBHIP_OK)
            throw std::runtime_error(orbhip_last_error());
        for (int i1 = 0; i1 < n1; i1++)
            if (m12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[m12[i1]];
        return nmatches;
    }

    // int SearchForTriangulation(KeyFrame *pKF1, KeyFrame* pKF2, cv::Mat F12, vector<cv::KeyPoint> &vMatchedKeys1,
    //                            vector<cv::KeyPoint> &vMatchedKeys2, vector<pair<size_t, size_t> > &vMatchedPairs)
    // (include/ORBmatcher.h:79, src/ORBmatcher.cc:863-1025): node pairing, Hamming distances, the epipolar test and
    // the rotation histogram run on the device; the three output vectors are filled as :1007-1022 do.
    template <class KeyFrameT>
    int SearchForTriangulation(KeyFrameT* pKF1, KeyFrameT* pKF2, cv::Mat F12, std::vector<cv::KeyPoint>& vMatchedKeys1,
                               std::vector<cv::KeyPoint>& vMatchedKeys2, std::vector<std::pair<size_t, size_t> >& vMatchedPairs) {
        auto vpMapPoints1 = pKF1->GetMapPointMatches();
        std::vector<cv::KeyPoint> vKeysUn1 = pKF1->GetKeyPointsUn();
        const std::vector<unsigned char> d1 = packRows(pKF1->GetDescriptors());
        const FlatFeatureVector fv1(pKF1->GetFeatureVector());
        auto vpMapPoints2 = pKF2->GetMapPointMatches();
        std::vector<cv::KeyPoint> vKeysUn2 = pKF2->GetKeyPointsUn();
        const std::vector<unsigned char> d2 = packRows(pKF2->GetDescriptors());
        const FlatFeatureVector fv2(pKF2->GetFeatureVector());
        const int n1 = (int)vKeysUn1.size(), n2 = (int)vKeysUn2.size();
        std::vector<unsigned char> free1(n1 ? n1 : 1), free2(n2 ? n2 : 1);
        for (int i = 0; i < n1; i++) free1[i] = !vpMapPoints1[i];
        for (int i = 0; i < n2; i++) free2[i] = !vpMapPoints2[i];
        const int nlevels = pKF2->GetScaleLevels();
        std::vector<float> sigma2(nlevels);
        for (int l = 0; l < nlevels; l++) sigma2[l] = pKF2->GetSigma2(l);
        float F[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) F[3 * i + j] = F12.template at<float>(i, j);
        std::vector<int> vMatches12(n1 ? n1 : 1, -1);
        int nmatches = 0;
        if (orbhip_search_for_triangulation(handle, (const OrbKeypoint*)vKeysUn1.data(), d1.data(), n1, free1.data(), fv1.nodes.data(),
                                            fv1.start.data(), fv1.feat.data(), fv1.count(), (const OrbKeypoint*)vKeysUn2.data(),
                                            d2.data(), n2, free2.data(), fv2.nodes.data(), fv2.start.data(), fv2.feat.data(),
                                            fv2.count(), F, sigma2.data(), nlevels, mbCheckOrientation, vMatches12.data(),
                                            &nmatches) != ORBHIP_OK)
            throw std::runtime_error(orbhip_last_error());
        vMatchedKeys1.clear();
        vMatchedKeys1.reserve(nmatches);
        vMatchedKeys2.clear();
        vMatchedKeys2.reserve(nmatches);
        vMatchedPairs.clear();
        vMatchedPairs.reserve(nmatches);
        for (size_t i = 0, iend = n1; i < iend; i++) {
            if (vMatches12[i] < 0) continue;
            vMatchedKeys1.push_back(vKeysUn1[i]);
            vMatchedKeys2.push_back(vKeysUn2[vMatches12[i]]);
            vMatchedPairs.push_back(std::make_pair(i, vMatches12[i]));
        }
        return nmatches;
    }

    // ---- the three per-frame / per-map-point steps around the matchers -----------------------------------
    // MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:191-256): the map point keeps the lines that
    // collect vDescriptors from its observations (:194-219) and the final mDescriptor = vDescriptors[BestIdx]
    // .clone() (:252-255); the N x N distances and the median choice (:221-250) are this call.
    int DistinctiveDescriptorIndex(const std::vector<cv::Mat>& vDescriptors) {
        const int N = (int)vDescriptors.size();
        if (N == 0) return -1;
        std::vector<unsigned char> d((size_t)N * 32);
        for (int i = 0; i < N; i++) CopyDescriptor(vDescriptors[i], &d[(size_t)i * 32]);
        const int start[2] = {0, N};
        int best = -1, median = 0;
        if (orbhip_distinctive_descriptors(handle, d.data(), start, 1, &best, &median) != ORBHIP_OK)
            throw std::runtime_error(orbhip_last_error());
        return best;
    }

    // Frame::UndistortKeyPoints (src/Frame.cc:290-320): mvKeysUn from mvKeys, mK and mDistCoef.
    template <class FrameT>
    void UndistortKeyPoints(FrameT& F) {
        const OrbCamera cam = {F.mK.template at<float>(0, 0), F.mK.template at<float>(1, 1), F.mK.template at<float>(0, 2),
                               F.mK.template at<float>(1, 2), F.mDistCoef.template at<float>(0), F.mDistCoef.template at<float>(1),
                               F.mDistCoef.template at<float>(2), F.mDistCoef.template at<float>(3)};
        F.mvKeysUn.resize(F.mvKeys.size());
        if (orbhip_undistort_keypoints(handle, (const OrbKeypoint*)F.mvKeys.data(), (int)F.mvKeys.size(), cam,
                                       (OrbKeypoint*)F.mvKeysUn.data()) != ORBHIP_OK)
            throw std::runtime_error(orbhip_last_error());
    }

    // Frame::ComputeBoW (src/Frame.cc:281-288) for the ORB vocabulary (TF-IDF weights, L1 scoring): the descent of
    // every descriptor runs on the device, the BowVector / FeatureVector are filled in feature order with the
    // containers' own addWeight / addFeature (TemplatedVocabulary.h:1150-1163) and normalised as transform() does
    // (:1193).  l1 is DBoW2::L1.
    template <class FrameT, class LNormT>
    void ComputeBoW(FrameT& F, const orbhip_vocabulary* voc, LNormT l1) {
        if (!F.mBowVec.empty()) return;
        const int n = F.mDescriptors.rows;
        const std::vector<unsigned char> d = packDescriptors(F);
        std::vector<int> word(n ? n : 1), node(n ? n : 1);
        std::vector<double> weight(n ? n : 1);
        if (orbhip_bow_transform(handle, voc, d.data(), n, 4, word.data(), node.data(), weight.data()) != ORBHIP_OK)
            throw std::runtime_error(orbhip_last_error());
        F.mBowVec.clear();
        F.mFeatVec.clear();
        for (int i = 0; i < n; i++)
            if (weight[i] > 0) {
                F.mBowVec.addWeight(word[i], weight[i]);
                F.mFeatVec.addFeature(node[i], i);
            }
        F.mBowVec.normalize(l1);
    }
};
}  // namespace ORB_SLAM
#endif
