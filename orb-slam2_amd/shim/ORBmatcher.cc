// Bodies of the drop-in ORB_SLAM::ORBmatcher (see ORBmatcher.h).  Every method has the same three steps: flatten
// what it needs from the caller's objects, one or two calls into liborbhip.so, write the results back.  No distance,
// projection or acceptance rule is evaluated here; the only decisions left on the host are the ones that change the
// caller's map (Fuse) or combine two searches (SearchBySim3), in the order the callers depend on.
#include "ORBmatcher.h"

#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "KeyFrame.h"
#include "MapPoint.h"
#include "../../include/orbhip.h"

namespace ORB_SLAM {

const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 20;
const int ORBmatcher::HISTO_LENGTH = 30;

namespace {

typedef unsigned char u8;

void check(int rc) {
    if (rc != ORBHIP_OK) throw std::runtime_error(orbhip_last_error());
}

struct ThreadHandle {
    orbhip_matcher* h;
    ThreadHandle() : h(0) {}
    ~ThreadHandle() {
        if (h) orbhip_matcher_destroy(h);
    }
};

OrbFrameBounds imageBounds() {
    const OrbFrameBounds b = {Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY};
    return b;
}

// rows of a CV_8U descriptor matrix, packed 32 bytes apart
void packRows(const cv::Mat& D, std::vector<u8>& out) {
    out.resize((size_t)(D.rows > 0 ? D.rows : 1) * 32);
    for (int i = 0; i < D.rows; i++) memcpy(&out[(size_t)i * 32], D.ptr<u8>(i), 32);
}
void packRow(const cv::Mat& d, u8* dst) { memcpy(dst, d.ptr<u8>(0), 32); }

void copy3(const cv::Mat& v, float* dst) {
    for (int k = 0; k < 3; k++) dst[k] = v.at<float>(k);
}

// a 4x4 pose or similarity as 16 floats, row-major
void flatten4x4(const cv::Mat& T, float out[16]) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[4 * i + j] = T.at<float>(i, j);
}
void poseOf(KeyFrame* pKF, float out[16]) {
    const cv::Mat R = pKF->GetRotation(), t = pKF->GetTranslation();
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) out[4 * i + j] = R.at<float>(i, j);
        out[4 * i + 3] = t.at<float>(i);
        out[12 + i] = 0.f;
    }
    out[15] = 1.f;
}

OrbProjection cameraOfFrame(const Frame& F) {
    float T[16];
    flatten4x4(F.mTcw, T);
    OrbProjection cam;
    check(orbhip_projection_from_pose(T, Frame::fx, Frame::fy, Frame::cx, Frame::cy, &cam));
    return cam;
}
OrbProjection cameraOfKeyFrame(KeyFrame* pKF) {
    float T[16];
    poseOf(pKF, T);
    OrbProjection cam;
    check(orbhip_projection_from_pose(T, pKF->fx, pKF->fy, pKF->cx, pKF->cy, &cam));
    return cam;
}
OrbProjection cameraOfSim3(KeyFrame* pKF, const cv::Mat& Scw) {
    float S[16];
    flatten4x4(Scw, S);
    OrbProjection cam;
    check(orbhip_projection_from_sim3(S, pKF->fx, pKF->fy, pKF->cx, pKF->cy, &cam));
    return cam;
}

// The map points of one call, flattened.  A slot that is dropped (skip) keeps zeros.
struct Points {
    std::vector<float> pos, normal, minDist, maxDist;
    std::vector<cv::KeyPoint> src;
    std::vector<u8> skip, desc;
    int n;
    explicit Points(int n_) : n(n_) {
        const size_t m = n > 0 ? n : 1;
        pos.assign(3 * m, 0.f);
        normal.assign(3 * m, 0.f);
        minDist.assign(m, 0.f);
        maxDist.assign(m, 0.f);
        src.resize(m);
        skip.assign(m, 1);
        desc.assign(32 * m, 0);
    }
    // geometry of a usable point; the descriptor is set separately (it is the map point's own in some methods and
    // the source frame's row in others)
    void take(int i, MapPoint* p) {
        skip[i] = 0;
        copy3(p->GetWorldPos(), &pos[3 * i]);
        copy3(p->GetNormal(), &normal[3 * i]);
        minDist[i] = p->GetMinDistanceInvariance();
        maxDist[i] = p->GetMaxDistanceInvariance();
    }
    void takePosition(int i, MapPoint* p) {
        skip[i] = 0;
        copy3(p->GetWorldPos(), &pos[3 * i]);
    }
    OrbPointSet view() const {
        static_assert(sizeof(cv::KeyPoint) == sizeof(OrbKeypoint), "cv::KeyPoint layout");
        const OrbPointSet s = {n, pos.data(), normal.data(), minDist.data(), maxDist.data(), (const OrbKeypoint*)src.data(),
                               skip.data(), desc.data()};
        return s;
    }
};

const OrbKeypoint* kp(const std::vector<cv::KeyPoint>& v) { return v.empty() ? 0 : (const OrbKeypoint*)&v[0]; }

// which features hold a map point on entry
std::vector<u8> occupied(const std::vector<MapPoint*>& v) {
    std::vector<u8> t(v.empty() ? 1 : v.size());
    for (size_t i = 0; i < v.size(); i++) t[i] = v[i] != 0;
    return t;
}

// DBoW2::FeatureVector (node id -> feature indices) as three arrays in map order
struct FlatFeatureVector {
    std::vector<int32_t> nodes, start, feat;
    template <class FV>
    explicit FlatFeatureVector(const FV& fv) {
        start.push_back(0);
        for (typename FV::const_iterator it = fv.begin(); it != fv.end(); ++it) {
            nodes.push_back((int32_t)it->first);
            feat.insert(feat.end(), it->second.begin(), it->second.end());
            start.push_back((int32_t)feat.size());
        }
        if (nodes.empty()) nodes.push_back(0);
        if (feat.empty()) feat.push_back(0);
    }
    int count() const { return (int)start.size() - 1; }
};

}  // namespace

orbhip_matcher* ORBmatcher::Accel() {
    static thread_local ThreadHandle th;
    if (!th.h) {
        const char* dev = getenv("ORBHIP_DEVICE");
        check(orbhip_matcher_create(&th.h, dev ? atoi(dev) : 0));
    }
    return th.h;
}

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {
    return orbhip_hamming_distance(a.ptr<int32_t>(), b.ptr<int32_t>());
}

// ---- projection matchers: points -> probes -> ordered search over the target's grid, all inside the library -----

int ORBmatcher::SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th) {
    const int np = (int)vpMapPoints.size(), n = (int)F.mvKeysUn.size();
    const size_t m = np > 0 ? np : 1;
    std::vector<float> proj(2 * m), viewCos(m);
    std::vector<int32_t> level(m);
    std::vector<u8> skip(m, 1), pdesc(32 * m);
    for (int i = 0; i < np; i++) {
        MapPoint* p = vpMapPoints[i];
        if (!p || !p->mbTrackInView || p->isBad()) continue;
        skip[i] = 0;
        proj[2 * i] = p->mTrackProjX;
        proj[2 * i + 1] = p->mTrackProjY;
        viewCos[i] = p->mTrackViewCos;
        level[i] = p->mnTrackScaleLevel;
        packRow(p->GetDescriptor(), &pdesc[(size_t)i * 32]);
    }
    std::vector<OrbProbe> probes(m);
    check(orbhip_probes_from_tracked(np, proj.data(), viewCos.data(), level.data(), skip.data(), th, F.mvScaleFactors.data(),
                                     (int)F.mvScaleFactors.size(), probes.data()));
    std::vector<u8> d2;
    packRows(F.mDescriptors, d2);
    const std::vector<u8> taken = occupied(F.mvpMapPoints);
    std::vector<int32_t> owner(n > 0 ? n : 1, -1);
    int nmatches = 0;
    check(orbhip_search_by_projection(Accel(), probes.data(), pdesc.data(), np, kp(F.mvKeysUn), d2.data(), n, imageBounds(),
                                      taken.data(), ORBHIP_PROJ_LOCAL_MAP, mfNNratio, TH_HIGH, 0, owner.data(), &nmatches));
    for (int i = 0; i < n; i++)
        if (owner[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[owner[i]];
    return nmatches;
}

int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, float th) {
    const int np = (int)LastFrame.mvpMapPoints.size(), n = (int)CurrentFrame.mvKeysUn.size();
    Points pts(np);
    for (int i = 0; i < np; i++) {
        MapPoint* p = LastFrame.mvpMapPoints[i];
        if (!p || LastFrame.mvbOutlier[i]) continue;
        pts.takePosition(i, p);
        pts.src[i] = LastFrame.mvKeysUn[i];            // the angle of the rotation check
        pts.src[i].octave = LastFrame.mvKeys[i].octave;  // the predicted octave
    }
    packRows(LastFrame.mDescriptors, pts.desc);
    pts.desc.resize(32 * (size_t)(np > 0 ? np : 1));
    std::vector<u8> d2;
    packRows(CurrentFrame.mDescriptors, d2);
    const std::vector<u8> taken = occupied(CurrentFrame.mvpMapPoints);
    std::vector<int32_t> owner(n > 0 ? n : 1, -1);
    int nmatches = 0;
    const OrbProjection cam = cameraOfFrame(CurrentFrame);
    const OrbPointSet set = pts.view();
    check(orbhip_project_and_search(Accel(), ORBHIP_PROBES_LAST_FRAME, &cam, CurrentFrame.mvScaleFactors.data(),
                                    (int)CurrentFrame.mvScaleFactors.size(), &set, th, 0.f, kp(CurrentFrame.mvKeysUn), d2.data(), n,
                                    imageBounds(), taken.data(), ORBHIP_PROJ_BEST, mfNNratio, TH_HIGH, mbCheckOrientation,
                                    owner.data(), &nmatches, 0));
    for (int i = 0; i < n; i++)
        if (owner[i] >= 0) CurrentFrame.mvpMapPoints[i] = LastFrame.mvpMapPoints[owner[i]];
    return nmatches;
}

int ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, float th,
                                   int ORBdist) {
    const std::vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
    const int np = (int)vpMPs.size(), n = (int)CurrentFrame.mvKeysUn.size();
    Points pts(np);
    for (int i = 0; i < np; i++) {
        MapPoint* p = vpMPs[i];
        if (!p || p->isBad() || sAlreadyFound.count(p)) continue;
        pts.takePosition(i, p);
        pts.minDist[i] = p->GetMinDistanceInvariance();
        pts.src[i] = pKF->GetKeyPointUn(i);
        packRow(p->GetDescriptor(), &pts.desc[(size_t)i * 32]);
    }
    std::vector<u8> d2;
    packRows(CurrentFrame.mDescriptors, d2);
    const std::vector<u8> taken = occupied(CurrentFrame.mvpMapPoints);
    std::vector<int32_t> owner(n > 0 ? n : 1, -1);
    int nmatches = 0;
    const OrbProjection cam = cameraOfFrame(CurrentFrame);
    OrbPointSet set = pts.view();
    set.normal = 0;
    set.max_dist = 0;
    check(orbhip_project_and_search(Accel(), ORBHIP_PROBES_KEYFRAME, &cam, CurrentFrame.mvScaleFactors.data(),
                                    CurrentFrame.mnScaleLevels, &set, th, 0.f, kp(CurrentFrame.mvKeysUn), d2.data(), n,
                                    imageBounds(), taken.data(), ORBHIP_PROJ_BEST, mfNNratio, ORBdist, mbCheckOrientation,
                                    owner.data(), &nmatches, 0));
    for (int i = 0; i < n; i++)
        if (owner[i] >= 0) CurrentFrame.mvpMapPoints[i] = vpMPs[owner[i]];
    return nmatches;
}

int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints,
                                   std::vector<MapPoint*>& vpMatched, int th) {
    const std::set<MapPoint*> found(vpMatched.begin(), vpMatched.end());  // may hold NULL: no point equals it
    const int np = (int)vpPoints.size();
    Points pts(np);
    for (int i = 0; i < np; i++) {
        MapPoint* p = vpPoints[i];
        if (p->isBad() || found.count(p)) continue;
        pts.take(i, p);
        packRow(p->GetDescriptor(), &pts.desc[(size_t)i * 32]);
    }
    const std::vector<cv::KeyPoint> keys = pKF->GetKeyPointsUn();
    const int n = (int)keys.size();
    std::vector<u8> d2;
    packRows(pKF->GetDescriptors(), d2);
    const std::vector<u8> taken = occupied(vpMatched);
    const std::vector<float> sf = pKF->GetScaleFactors();
    std::vector<int32_t> owner(n > 0 ? n : 1, -1);
    int nmatches = 0;
    const OrbProjection cam = cameraOfSim3(pKF, Scw);
    const OrbPointSet set = pts.view();
    check(orbhip_project_and_search(Accel(), ORBHIP_PROBES_FUSE, &cam, sf.data(), pKF->GetScaleLevels(), &set, (float)th, 0.f,
                                    kp(keys), d2.data(), n, imageBounds(), taken.data(), ORBHIP_PROJ_BEST, mfNNratio, TH_LOW, 0,
                                    owner.data(), &nmatches, 0));
    for (int i = 0; i < n; i++)
        if (owner[i] >= 0) vpMatched[i] = vpPoints[owner[i]];
    return nmatches;
}

int ORBmatcher::SearchByProjection(Frame& F1, Frame& F2, int windowSize, std::vector<MapPoint*>& vpMapPointMatches2) {
    vpMapPointMatches2 = F2.mvpMapPoints;
    const std::set<MapPoint*> found(vpMapPointMatches2.begin(), vpMapPointMatches2.end());
    const int np = (int)F1.mvpMapPoints.size(), n = (int)F2.mvKeysUn.size();
    Points pts(np);
    for (int i = 0; i < np; i++) {
        MapPoint* p = F1.mvpMapPoints[i];
        if (!p || p->isBad() || found.count(p)) continue;
        pts.takePosition(i, p);
        pts.src[i] = F1.mvKeysUn[i];
    }
    packRows(F1.mDescriptors, pts.desc);
    pts.desc.resize(32 * (size_t)(np > 0 ? np : 1));
    std::vector<u8> d2;
    packRows(F2.mDescriptors, d2);
    const std::vector<u8> taken = occupied(vpMapPointMatches2);
    const float one = 1.f;  // this method reads no scale factor; the library wants a table of at least one level
    std::vector<int32_t> owner(n > 0 ? n : 1, -1);
    int nmatches = 0;
    const OrbProjection cam = cameraOfFrame(F2);
    const OrbPointSet set = pts.view();
    check(orbhip_project_and_search(Accel(), ORBHIP_PROBES_WINDOW, &cam, F2.mvScaleFactors.empty() ? &one : F2.mvScaleFactors.data(),
                                    F2.mvScaleFactors.empty() ? 1 : (int)F2.mvScaleFactors.size(), &set, (float)windowSize, 0.f,
                                    kp(F2.mvKeysUn), d2.data(), n, imageBounds(), taken.data(), ORBHIP_PROJ_RATIO, mfNNratio,
                                    TH_HIGH, 0, owner.data(), &nmatches, 0));
    for (int i = 0; i < n; i++)
        if (owner[i] >= 0) vpMapPointMatches2[i] = F1.mvpMapPoints[owner[i]];
    return nmatches;
}

// ---- windowed matchers --------------------------------------------------------------------------------------------

int ORBmatcher::WindowSearch(Frame& F1, Frame& F2, int windowSize, std::vector<MapPoint*>& vpMapPointMatches2, int minOctave,
                             int maxOctave) {
    const int n1 = (int)F1.mvpMapPoints.size(), n2 = (int)F2.mvKeysUn.size();
    vpMapPointMatches2.assign(F2.mvpMapPoints.size(), static_cast<MapPoint*>(0));
    std::vector<u8> usable(n1 > 0 ? n1 : 1, 0), d1, d2;
    for (int i = 0; i < n1; i++) usable[i] = F1.mvpMapPoints[i] && !F1.mvpMapPoints[i]->isBad();
    packRows(F1.mDescriptors, d1);
    packRows(F2.mDescriptors, d2);
    std::vector<int32_t> from1(n2 > 0 ? n2 : 1, -1);
    int nmatches = 0;
    check(orbhip_window_search(Accel(), kp(F1.mvKeysUn), d1.data(), n1, usable.data(), kp(F2.mvKeysUn), d2.data(), n2, imageBounds(),
                               windowSize, minOctave, maxOctave, mfNNratio, mbCheckOrientation, from1.data(), &nmatches));
    for (int i2 = 0; i2 < n2; i2++)
        if (from1[i2] >= 0) vpMapPointMatches2[i2] = F1.mvpMapPoints[from1[i2]];
    return nmatches;
}

int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                                        int windowSize) {
    static_assert(sizeof(cv::Point2f) == 2 * sizeof(float), "cv::Point2f layout");
    const int n1 = (int)F1.mvKeysUn.size(), n2 = (int)F2.mvKeysUn.size();
    vnMatches12.assign(n1, -1);
    std::vector<u8> d1, d2;
    packRows(F1.mDescriptors, d1);
    packRows(F2.mDescriptors, d2);
    int nmatches = 0;
    check(orbhip_search_for_initialization(Accel(), kp(F1.mvKeysUn), d1.data(), n1, kp(F2.mvKeysUn), d2.data(), n2, imageBounds(),
                                           vbPrevMatched.empty() ? 0 : &vbPrevMatched[0].x, windowSize, mfNNratio,
                                           mbCheckOrientation, vnMatches12.empty() ? 0 : &vnMatches12[0], &nmatches));
    return nmatches;
}

// ---- vocabulary-node matchers -------------------------------------------------------------------------------------

int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) {
    const std::vector<MapPoint*> held = pKF->GetMapPointMatches();
    vpMapPointMatches.assign(F.mvpMapPoints.size(), static_cast<MapPoint*>(0));
    const FlatFeatureVector fvKF(pKF->GetFeatureVector()), fvF(F.mFeatVec);
    const std::vector<cv::KeyPoint> keysKF = pKF->GetKeyPointsUn();
    const int nKF = (int)keysKF.size(), nF = (int)F.mvKeys.size();
    std::vector<u8> dKF, dF, usable(nKF > 0 ? nKF : 1, 0);
    packRows(pKF->GetDescriptors(), dKF);
    packRows(F.mDescriptors, dF);
    // a feature index beyond the keyframe's map points counts as a feature without one
    for (int i = 0; i < nKF && i < (int)held.size(); i++) usable[i] = held[i] && !held[i]->isBad();
    std::vector<int32_t> m12(nKF > 0 ? nKF : 1, -1), m21(nF > 0 ? nF : 1, -1);
    int nmatches = 0;
    check(orbhip_search_by_bow(Accel(), kp(keysKF), dKF.data(), nKF, usable.data(), fvKF.nodes.data(), fvKF.start.data(),
                               fvKF.feat.data(), fvKF.count(), kp(F.mvKeys), dF.data(), nF, 0, fvF.nodes.data(), fvF.start.data(),
                               fvF.feat.data(), fvF.count(), mfNNratio, mbCheckOrientation, 1, m12.data(), m21.data(), &nmatches));
    for (int iF = 0; iF < nF; iF++)
        if (m21[iF] >= 0) vpMapPointMatches[iF] = held[m21[iF]];
    return nmatches;
}

int ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) {
    KeyFrame* kf[2] = {pKF1, pKF2};
    std::vector<cv::KeyPoint> keys[2];
    std::vector<MapPoint*> held[2];
    std::vector<u8> desc[2], usable[2];
    for (int s = 0; s < 2; s++) {
        keys[s] = kf[s]->GetKeyPointsUn();
        held[s] = kf[s]->GetMapPointMatches();
        packRows(kf[s]->GetDescriptors(), desc[s]);
        usable[s].assign(keys[s].empty() ? 1 : keys[s].size(), 0);
        for (size_t i = 0; i < keys[s].size() && i < held[s].size(); i++) usable[s][i] = held[s][i] && !held[s][i]->isBad();
    }
    const FlatFeatureVector fv1(pKF1->GetFeatureVector()), fv2(pKF2->GetFeatureVector());
    vpMatches12.assign(held[0].size(), static_cast<MapPoint*>(0));
    const int n1 = (int)keys[0].size(), n2 = (int)keys[1].size();
    std::vector<int32_t> m12(n1 > 0 ? n1 : 1, -1), m21(n2 > 0 ? n2 : 1, -1);
    int nmatches = 0;
    check(orbhip_search_by_bow(Accel(), kp(keys[0]), desc[0].data(), n1, usable[0].data(), fv1.nodes.data(), fv1.start.data(),
                               fv1.feat.data(), fv1.count(), kp(keys[1]), desc[1].data(), n2, usable[1].data(), fv2.nodes.data(),
                               fv2.start.data(), fv2.feat.data(), fv2.count(), mfNNratio, mbCheckOrientation, 0, m12.data(),
                               m21.data(), &nmatches));
    for (int i1 = 0; i1 < n1; i1++)
        if (m12[i1] >= 0) vpMatches12[i1] = held[1][m12[i1]];
    return nmatches;
}

int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<cv::KeyPoint>& vMatchedKeys1,
                                       std::vector<cv::KeyPoint>& vMatchedKeys2,
                                       std::vector<std::pair<size_t, size_t> >& vMatchedPairs) {
    KeyFrame* kf[2] = {pKF1, pKF2};
    std::vector<cv::KeyPoint> keys[2];
    std::vector<u8> desc[2], unmatched[2];
    for (int s = 0; s < 2; s++) {
        keys[s] = kf[s]->GetKeyPointsUn();
        packRows(kf[s]->GetDescriptors(), desc[s]);
        const std::vector<MapPoint*> held = kf[s]->GetMapPointMatches();
        unmatched[s].assign(keys[s].empty() ? 1 : keys[s].size(), 0);
        for (size_t i = 0; i < keys[s].size(); i++) unmatched[s][i] = !held[i];
    }
    const FlatFeatureVector fv1(pKF1->GetFeatureVector()), fv2(pKF2->GetFeatureVector());
    const int nlevels = pKF2->GetScaleLevels();
    std::vector<float> sigma2(nlevels);
    for (int l = 0; l < nlevels; l++) sigma2[l] = pKF2->GetSigma2(l);
    float F[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) F[3 * i + j] = F12.at<float>(i, j);
    const int n1 = (int)keys[0].size(), n2 = (int)keys[1].size();
    std::vector<int32_t> m12(n1 > 0 ? n1 : 1, -1);
    int nmatches = 0;
    check(orbhip_search_for_triangulation(Accel(), kp(keys[0]), desc[0].data(), n1, unmatched[0].data(), fv1.nodes.data(),
                                          fv1.start.data(), fv1.feat.data(), fv1.count(), kp(keys[1]), desc[1].data(), n2,
                                          unmatched[1].data(), fv2.nodes.data(), fv2.start.data(), fv2.feat.data(), fv2.count(), F,
                                          sigma2.data(), nlevels, mbCheckOrientation, m12.data(), &nmatches));
    vMatchedKeys1.clear();
    vMatchedKeys2.clear();
    vMatchedPairs.clear();
    vMatchedKeys1.reserve(nmatches);
    vMatchedKeys2.reserve(nmatches);
    vMatchedPairs.reserve(nmatches);
    for (int i1 = 0; i1 < n1; i1++) {
        if (m12[i1] < 0) continue;
        vMatchedKeys1.push_back(keys[0][i1]);
        vMatchedKeys2.push_back(keys[1][m12[i1]]);
        vMatchedPairs.push_back(std::make_pair((size_t)i1, (size_t)m12[i1]));
    }
    return nmatches;
}

// ---- independent per-point searches: Fuse and SearchBySim3 --------------------------------------------------------

namespace {

// Best feature of one keyframe for every point of a set (projection with `cam`, best-in-radius search), with the
// keyframe's features flattened once.
struct KeyFrameSearch {
    std::vector<cv::KeyPoint> keys;
    std::vector<u8> desc;
    std::vector<float> sf;
    int nlevels;
    explicit KeyFrameSearch(KeyFrame* pKF) : keys(pKF->GetKeyPointsUn()), sf(pKF->GetScaleFactors()), nlevels(pKF->GetScaleLevels()) {
        packRows(pKF->GetDescriptors(), desc);
    }
    void run(int kind, const OrbProjection& cam, const OrbProjection* camTo, const Points& pts, float th, std::vector<int32_t>& idx,
             std::vector<int32_t>& dist) const {
        idx.assign(pts.n > 0 ? pts.n : 1, -1);
        dist.assign(pts.n > 0 ? pts.n : 1, INT_MAX);
        const OrbPointSet set = pts.view();
        check(orbhip_project_and_best_in_radius(ORBmatcher::Accel(), kind, &cam, camTo, sf.data(), nlevels, &set, th, kp(keys),
                                                desc.data(), (int)keys.size(), imageBounds(), idx.data(), dist.data(), 0));
    }
};

void describe(Points& pts, int i, MapPoint* p) {
    pts.take(i, p);
    packRow(p->GetDescriptor(), &pts.desc[(size_t)i * 32]);
}

// Shared by the two forms of Fuse.  Every listed point is searched in one call; the list is then walked in order,
// because fusing changes what later entries see (a point can become bad, or enter the keyframe).  Only a point that
// is listed a second time can have a different search result by then (a Replace in between may have given it another
// descriptor), so it is searched again at its turn.  `drop` and `merge` carry what differs between the two forms.
template <class Drop, class Merge>
int fuseWalk(KeyFrame* pKF, const std::vector<MapPoint*>& list, const OrbProjection& cam, float th, int thLow, Drop drop,
             Merge merge) {
    const KeyFrameSearch search(pKF);
    const int np = (int)list.size();
    Points pts(np);
    for (int i = 0; i < np; i++)
        if (list[i]) describe(pts, i, list[i]);
    std::vector<int32_t> idx, dist;
    search.run(ORBHIP_PROBES_FUSE, cam, 0, pts, th, idx, dist);
    std::set<MapPoint*> visited;
    int nFused = 0;
    for (int i = 0; i < np; i++) {
        MapPoint* p = list[i];
        if (!p || drop(p)) continue;
        int feature = idx[i], d = dist[i];
        if (!visited.insert(p).second) {
            Points again(1);
            describe(again, 0, p);
            std::vector<int32_t> i1, d1;
            search.run(ORBHIP_PROBES_FUSE, cam, 0, again, th, i1, d1);
            feature = i1[0];
            d = d1[0];
        }
        if (d > thLow) continue;
        merge(p, feature);
        nFused++;
    }
    return nFused;
}

struct DropIfInKeyFrame {
    KeyFrame* kf;
    bool operator()(MapPoint* p) const { return p->isBad() || p->IsInKeyFrame(kf); }
};
struct DropIfListed {
    const std::set<MapPoint*>* listed;
    bool operator()(MapPoint* p) const { return p->isBad() || listed->count(p) != 0; }
};
// the keyframe's own point survives (LocalMapping) / the candidate survives (loop closing)
struct MergeIntoKeyFramePoint {
    KeyFrame* kf;
    void operator()(MapPoint* p, int feature) const {
        MapPoint* own = kf->GetMapPoint(feature);
        if (!own) {
            p->AddObservation(kf, feature);
            kf->AddMapPoint(p, feature);
        } else if (!own->isBad()) {
            p->Replace(own);
        }
    }
};
struct MergeIntoCandidate {
    KeyFrame* kf;
    void operator()(MapPoint* p, int feature) const {
        MapPoint* own = kf->GetMapPoint(feature);
        if (!own) {
            p->AddObservation(kf, feature);
            kf->AddMapPoint(p, feature);
        } else if (!own->isBad()) {
            own->Replace(p);
        }
    }
};

}  // namespace

int ORBmatcher::Fuse(KeyFrame* pKF, std::vector<MapPoint*>& vpMapPoints, float th) {
    const DropIfInKeyFrame drop = {pKF};
    const MergeIntoKeyFramePoint merge = {pKF};
    return fuseWalk(pKF, vpMapPoints, cameraOfKeyFrame(pKF), th, TH_LOW, drop, merge);
}

int ORBmatcher::Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th) {
    const std::set<MapPoint*> inKeyFrame = pKF->GetMapPoints();  // as they are on entry
    const DropIfListed drop = {&inKeyFrame};
    const MergeIntoCandidate merge = {pKF};
    return fuseWalk(pKF, vpPoints, cameraOfSim3(pKF, Scw), th, TH_LOW, drop, merge);
}

int ORBmatcher::SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12,
                             const cv::Mat& t12, float th) {
    KeyFrame* kf[2] = {pKF1, pKF2};
    std::vector<MapPoint*> held[2] = {pKF1->GetMapPointMatches(), pKF2->GetMapPointMatches()};
    const int n[2] = {(int)held[0].size(), (int)held[1].size()};
    // what vpMatches12 holds on entry is excluded on both sides
    std::vector<bool> matched[2] = {std::vector<bool>(n[0], false), std::vector<bool>(n[1], false)};
    for (int i = 0; i < n[0]; i++) {
        MapPoint* p = vpMatches12[i];
        if (!p) continue;
        matched[0][i] = true;
        const int partner = p->GetIndexInKeyFrame(pKF2);
        if (partner >= 0 && partner < n[1]) matched[1][partner] = true;
    }
    float R[9], t[3];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) R[3 * i + j] = R12.at<float>(i, j);
        t[i] = t12.at<float>(i);
    }
    // both directions are projected with keyframe 1's intrinsics
    OrbProjection into[2];  // into[0]: camera 2 -> camera 1, into[1]: camera 1 -> camera 2
    check(orbhip_projection_sim3_pair(s12, R, t, pKF1->fx, pKF1->fy, pKF1->cx, pKF1->cy, &into[1], &into[0]));
    std::vector<int32_t> best[2], dist[2];
    for (int s = 0; s < 2; s++) {  // points of keyframe s searched among the features of keyframe 1 - s
        Points pts(n[s]);
        for (int i = 0; i < n[s]; i++) {
            MapPoint* p = held[s][i];
            if (!p || matched[s][i] || p->isBad()) continue;
            describe(pts, i, p);
        }
        const KeyFrameSearch target(kf[1 - s]);
        target.run(ORBHIP_PROBES_SIM3, cameraOfKeyFrame(kf[s]), &into[1 - s], pts, th, best[s], dist[s]);
        for (int i = 0; i < n[s]; i++)
            if (dist[s][i] > TH_HIGH) best[s][i] = -1;
    }
    // a pair counts when each side chose the other
    int nFound = 0;
    for (int i1 = 0; i1 < n[0]; i1++) {
        const int i2 = best[0][i1];
        if (i2 < 0 || best[1][i2] != i1) continue;
        vpMatches12[i1] = held[1][i2];
        nFound++;
    }
    return nFound;
}

}  // namespace ORB_SLAM
