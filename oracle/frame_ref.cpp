// CPU ORACLE (test infrastructure only; never linked into or loaded by the product library).
// PARITY UNPINNED: the reference ships no tests or fixtures for this path and cannot be built in
// this image, so these restatements are pinned only by their own known-answer tests.
//
// Literal restatement of the feature grid of ORB_SLAM::Frame and of the two windowed matchers of
// ORB_SLAM::ORBmatcher that take their candidates from it:
//   Frame grid            src/Frame.cc:78-79,110-124,268-278, include/Frame.h:37-38,90-92
//   GetFeaturesInArea     src/Frame.cc:201-266
//   WindowSearch          src/ORBmatcher.cc:420-527
//   SearchForInitialization  src/ORBmatcher.cc:609-724
// of the five projection matchers (pose arithmetic included, see the notes above them):
//   SearchByProjection    src/ORBmatcher.cc:52-128, :297-418, :530-605, :1518-1631, :1633-1757
//   KeyFrame::GetFeaturesInArea / IsInImage  src/KeyFrame.cc:638-683
// and of the descriptor choice of a map point:
//   MapPoint::ComputeDistinctiveDescriptors  src/MapPoint.cc:191-256
// Arithmetic follows the reference expression by expression in fp32 (this file is compiled with
// -ffp-contract=off).  Unqualified floor/ceil/round/abs in src/Frame.cc resolve to the float
// overloads of namespace std (pulled in through Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h).
#include "orb_ref.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

const int GRID_ROWS = 48;  // include/Frame.h:37
const int GRID_COLS = 64;  // include/Frame.h:38
const int HISTO_LENGTH = 30;
const int TH_HIGH = 100, TH_LOW = 20;  // src/ORBmatcher.cc:41-44

struct Kp {  // field-for-field the 28-byte keypoint record
    float x, y, size, angle, response;
    int32_t octave, class_id;
};

struct FrameRef {
    const Kp* kp;
    const uint8_t* desc;
    int n;
    int mnMinX, mnMinY, mnMaxX, mnMaxY;
    float mfGridElementWidthInv, mfGridElementHeightInv;
    std::vector<size_t> mGrid[GRID_COLS][GRID_ROWS];

    FrameRef(const void* kp_, const uint8_t* desc_, int n_, const int32_t bounds[4])
        : kp((const Kp*)kp_), desc(desc_), n(n_), mnMinX(bounds[0]), mnMinY(bounds[1]), mnMaxX(bounds[2]), mnMaxY(bounds[3]) {
        // src/Frame.cc:78-79
        mfGridElementWidthInv = static_cast<float>(GRID_COLS) / static_cast<float>(mnMaxX - mnMinX);
        mfGridElementHeightInv = static_cast<float>(GRID_ROWS) / static_cast<float>(mnMaxY - mnMinY);
        // src/Frame.cc:117-124
        for (size_t i = 0; i < (size_t)n; i++) {
            int nGridPosX, nGridPosY;
            if (PosInGrid(kp[i], nGridPosX, nGridPosY)) mGrid[nGridPosX][nGridPosY].push_back(i);
        }
    }

    // src/Frame.cc:268-278
    bool PosInGrid(const Kp& k, int& posX, int& posY) const {
        posX = std::round((k.x - mnMinX) * mfGridElementWidthInv);
        posY = std::round((k.y - mnMinY) * mfGridElementHeightInv);
        if (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) return false;
        return true;
    }

    // src/Frame.cc:201-266
    std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, int minLevel, int maxLevel) const {
        std::vector<size_t> vIndices;
        int nMinCellX = std::floor((x - mnMinX - r) * mfGridElementWidthInv);
        nMinCellX = std::max(0, nMinCellX);
        if (nMinCellX >= GRID_COLS) return vIndices;
        int nMaxCellX = std::ceil((x - mnMinX + r) * mfGridElementWidthInv);
        nMaxCellX = std::min(GRID_COLS - 1, nMaxCellX);
        if (nMaxCellX < 0) return vIndices;
        int nMinCellY = std::floor((y - mnMinY - r) * mfGridElementHeightInv);
        nMinCellY = std::max(0, nMinCellY);
        if (nMinCellY >= GRID_ROWS) return vIndices;
        int nMaxCellY = std::ceil((y - mnMinY + r) * mfGridElementHeightInv);
        nMaxCellY = std::min(GRID_ROWS - 1, nMaxCellY);
        if (nMaxCellY < 0) return vIndices;
        bool bCheckLevels = true;
        bool bSameLevel = false;
        if (minLevel == -1 && maxLevel == -1)
            bCheckLevels = false;
        else if (minLevel == maxLevel)
            bSameLevel = true;
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const std::vector<size_t>& vCell = mGrid[ix][iy];
                if (vCell.empty()) continue;
                for (size_t j = 0, jend = vCell.size(); j < jend; j++) {
                    const Kp& kpUn = kp[vCell[j]];
                    if (bCheckLevels && !bSameLevel) {
                        if (kpUn.octave < minLevel || kpUn.octave > maxLevel) continue;
                    } else if (bSameLevel) {
                        if (kpUn.octave != minLevel) continue;
                    }
                    if (std::abs(kpUn.x - x) > r || std::abs(kpUn.y - y) > r) continue;
                    vIndices.push_back(vCell[j]);
                }
            }
        }
        return vIndices;
    }
};

void three_maxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {
    int32_t sizes[HISTO_LENGTH], ind[3];
    for (int i = 0; i < L; i++) sizes[i] = (int)histo[i].size();
    orbref_three_maxima(sizes, L, ind);
    ind1 = ind[0];
    ind2 = ind[1];
    ind3 = ind[2];
}

}  // namespace

extern "C" {

void orbref_frame_grid(const void* kp, int n, const int32_t bounds[4], int32_t* cell_start, int32_t* items) {
    FrameRef F(kp, nullptr, n, bounds);
    int o = 0;
    for (int ix = 0; ix < GRID_COLS; ix++)
        for (int iy = 0; iy < GRID_ROWS; iy++) {
            cell_start[ix * GRID_ROWS + iy] = o;
            for (size_t v : F.mGrid[ix][iy]) items[o++] = (int32_t)v;
        }
    cell_start[GRID_COLS * GRID_ROWS] = o;
}

int orbref_features_in_area(const void* kp, int n, const int32_t bounds[4], float x, float y, float r, int min_level,
                            int max_level, int32_t* out, int cap) {
    FrameRef F(kp, nullptr, n, bounds);
    std::vector<size_t> v = F.GetFeaturesInArea(x, y, r, min_level, max_level);
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = (int32_t)v[i];
    return (int)v.size();
}

// src/ORBmatcher.cc:609-724
int orbref_search_for_initialization(const void* kp1, const uint8_t* desc1, int n1, const void* kp2, const uint8_t* desc2,
                                     int n2, const int32_t bounds[4], float* prev_matched_xy, int windowSize,
                                     float mfNNratio, int mbCheckOrientation, int32_t* vnMatches12) {
    FrameRef F1(kp1, desc1, n1, bounds), F2(kp2, desc2, n2, bounds);
    int nmatches = 0;
    for (int i = 0; i < n1; i++) vnMatches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> vMatchedDistance(n2, INT_MAX);
    std::vector<int> vnMatches21(n2, -1);
    for (size_t i1 = 0, iend1 = n1; i1 < iend1; i1++) {
        const Kp& kp1i = F1.kp[i1];
        int level1 = kp1i.octave;
        if (level1 > 0) continue;
        std::vector<size_t> vIndices2 =
            F2.GetFeaturesInArea(prev_matched_xy[2 * i1], prev_matched_xy[2 * i1 + 1], windowSize, level1, level1);
        if (vIndices2.empty()) continue;
        const uint8_t* d1 = desc1 + i1 * 32;
        int bestDist = INT_MAX;
        int bestDist2 = INT_MAX;
        int bestIdx2 = -1;
        for (size_t i2 : vIndices2) {
            int dist = orbref_hamming(d1, desc2 + i2 * 32);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestIdx2 = (int)i2;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * mfNNratio) {
                if (vnMatches21[bestIdx2] >= 0) {
                    vnMatches12[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = (int)i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (mbCheckOrientation) {
                    float rot = F1.kp[i1].angle - F2.kp[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back((int)i1);
                }
            }
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                int idx1 = rotHist[i][j];
                if (vnMatches12[idx1] >= 0) {
                    vnMatches12[idx1] = -1;
                    nmatches--;
                }
            }
        }
    }
    // "Update prev matched"
    for (int i1 = 0; i1 < n1; i1++)
        if (vnMatches12[i1] >= 0) {
            prev_matched_xy[2 * i1] = F2.kp[vnMatches12[i1]].x;
            prev_matched_xy[2 * i1 + 1] = F2.kp[vnMatches12[i1]].y;
        }
    return nmatches;
}

// src/ORBmatcher.cc:420-527.  valid1[i] != 0 stands for "F1.mvpMapPoints[i] is a good map point";
// vnMatches21[i2] = index in F1 whose map point was assigned to feature i2 of F2, else -1
// (vpMapPointMatches2[i2] = F1.mvpMapPoints[vnMatches21[i2]]).
int orbref_window_search(const void* kp1, const uint8_t* desc1, int n1, const uint8_t* valid1, const void* kp2,
                         const uint8_t* desc2, int n2, const int32_t bounds[4], int windowSize, int minScaleLevel,
                         int maxScaleLevel, float mfNNratio, int mbCheckOrientation, int32_t* vnMatches21) {
    FrameRef F1(kp1, desc1, n1, bounds), F2(kp2, desc2, n2, bounds);
    int nmatches = 0;
    for (int i = 0; i < n2; i++) vnMatches21[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    const bool bMinLevel = minScaleLevel > 0;
    const bool bMaxLevel = maxScaleLevel < INT_MAX;
    for (size_t i1 = 0, iend1 = n1; i1 < iend1; i1++) {
        if (!valid1[i1]) continue;
        const Kp& kp1i = F1.kp[i1];
        int level1 = kp1i.octave;
        if (bMinLevel)
            if (level1 < minScaleLevel) continue;
        if (bMaxLevel)
            if (level1 > maxScaleLevel) continue;
        std::vector<size_t> vIndices2 = F2.GetFeaturesInArea(kp1i.x, kp1i.y, windowSize, level1, level1);
        if (vIndices2.empty()) continue;
        const uint8_t* d1 = desc1 + i1 * 32;
        int bestDist = INT_MAX;
        int bestDist2 = INT_MAX;
        int bestIdx2 = -1;
        for (size_t i2 : vIndices2) {
            if (vnMatches21[i2] >= 0) continue;  // "if(vpMapPointMatches2[i2]) continue;"
            int dist = orbref_hamming(d1, desc2 + i2 * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestIdx2 = (int)i2;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        if (bestDist <= bestDist2 * mfNNratio && bestDist <= TH_HIGH) {
            vnMatches21[bestIdx2] = (int)i1;
            nmatches++;
            float rot = F1.kp[i1].angle - F2.kp[bestIdx2].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = std::round(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            rotHist[bin].push_back(bestIdx2);
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i != ind1 && i != ind2 && i != ind3) {
                for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                    vnMatches21[rotHist[i][j]] = -1;
                    nmatches--;
                }
            }
        }
    }
    return nmatches;
}

// src/MapPoint.cc:221-250 for one map point with N observed descriptors (rows of 32 bytes):
// returns BestIdx, *best_median = BestMedian.  N == 0 returns -1 (the reference returns early).
int orbref_distinctive_descriptor(const uint8_t* desc, int N, int32_t* best_median) {
    if (best_median) *best_median = INT_MAX;
    if (N <= 0) return -1;
    std::vector<float> Distances((size_t)N * N);
    for (size_t i = 0; i < (size_t)N; i++) {
        Distances[i * N + i] = 0;
        for (size_t j = i + 1; j < (size_t)N; j++) {
            int distij = orbref_hamming(desc + i * 32, desc + j * 32);
            Distances[i * N + j] = distij;
            Distances[j * N + i] = distij;
        }
    }
    int BestMedian = INT_MAX;
    int BestIdx = 0;
    for (size_t i = 0; i < (size_t)N; i++) {
        std::vector<int> vDists(Distances.begin() + i * N, Distances.begin() + (i + 1) * N);
        std::sort(vDists.begin(), vDists.end());
        int median = vDists[0.5 * (N - 1)];
        if (median < BestMedian) {
            BestMedian = median;
            BestIdx = (int)i;
        }
    }
    if (best_median) *best_median = BestMedian;
    return BestIdx;
}


// ---------------------------------------------------------------------------------------------
// The five ORBmatcher::SearchByProjection overloads.  Map points, poses and cameras arrive as plain
// arrays; a `skip` flag per map point stands for the pointer tests the reference makes before it
// projects (NULL, isBad(), membership in an already-found set).  Every function also returns the
// probe it built per map point (projection, radius, level range, angle, valid) so that a test can
// hand the same probes to the device search.
//
// cv::Mat expressions are restated under these assumptions about OpenCV 2.4 (no OpenCV in this image
// to confirm them — part of why this file is PARITY UNPINNED):
//   A*x + t   (3x3 * 3x1 + 3x1, CV_32F)  one cv::gemm call through its small-matrix path: every row
//             is t0 = a0*x0 + a1*x1 + a2*x2 in float, left to right, then (float)(t0*1.0 + t*1.0)
//             with the scale factors as doubles;
//   -A.t()*t  cv::gemm with GEMM_1_T and alpha = -1 through the general path: double accumulation
//             over k = 0,1,2, times alpha, one rounding to float;
//   cv::norm  double accumulation of squares, sqrt in double;   Mat::dot  double accumulation;
//   M / s     M.convertTo(.., 1./s): elementwise float multiply by (float)(1.0/s).
namespace {

struct ProbeRec {  // same layout as the device probe record
    float u, v, r;
    int32_t min_level, max_level;
    float angle;
    int32_t valid;
    int32_t reserved;
};

void clear_probe(ProbeRec* p) {
    if (p) memset(p, 0, sizeof(*p));
}

// Rcw (rows of the 4x4 row-major Tcw) * x + tcw
void rigid(const float* T, const float* x, float out[3]) {
    for (int i = 0; i < 3; i++) {
        const float* a = T + 4 * i;
        float t0 = a[0] * x[0] + a[1] * x[1] + a[2] * x[2];
        out[i] = (float)(t0 * 1.0 + a[3] * 1.0);
    }
}

// Ow = -Rcw.t()*tcw
void centre(const float* R /*9, row-major*/, const float* t, float Ow[3]) {
    for (int i = 0; i < 3; i++) {
        double s0 = 0;
        for (int k = 0; k < 3; k++) s0 += (double)R[3 * k + i] * (double)t[k];
        Ow[i] = (float)(s0 * -1.0);
    }
}

double norm3(const float* v) {
    double s0 = 0;
    for (int i = 0; i < 3; i++) s0 += (double)v[i] * v[i];
    return std::sqrt(s0);
}

double dot3(const float* a, const float* b) {
    double r = 0;
    for (int i = 0; i < 3; i++) r += (double)a[i] * b[i];
    return r;
}

}  // namespace

// SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th), src/ORBmatcher.cc:52-128.
// taken[idx] != 0: F.mvpMapPoints[idx] is set on entry.  match_of_feature[idx] = iMP assigned by this call, else -1.
int orbref_sbp_local_map(const void* kp, const uint8_t* desc, int n, const int32_t bounds[4], const float* mvScaleFactors,
                         const uint8_t* taken, int nmp, const uint8_t* mbTrackInView, const uint8_t* bad,
                         const int32_t* mnTrackScaleLevel, const float* mTrackViewCos, const float* mTrackProjX,
                         const float* mTrackProjY, const uint8_t* mp_desc, float th, float mfNNratio,
                         int32_t* match_of_feature, void* probes_out) {
    FrameRef F(kp, desc, n, bounds);
    ProbeRec* pr = (ProbeRec*)probes_out;
    std::vector<char> has(n);
    for (int i = 0; i < n; i++) {
        has[i] = taken && taken[i];
        match_of_feature[i] = -1;
    }
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    for (size_t iMP = 0; iMP < (size_t)nmp; iMP++) {
        clear_probe(pr ? pr + iMP : nullptr);
        if (!mbTrackInView[iMP]) continue;
        if (bad[iMP]) continue;
        const int& nPredictedLevel = mnTrackScaleLevel[iMP];
        float r = mTrackViewCos[iMP] > 0.998 ? 2.5 : 4.0;  // RadiusByViewingCos, :130-136
        if (bFactor) r *= th;
        const float rr = r * mvScaleFactors[nPredictedLevel];
        if (pr) {
            ProbeRec& q = pr[iMP];
            q.u = mTrackProjX[iMP];
            q.v = mTrackProjY[iMP];
            q.r = rr;
            q.min_level = nPredictedLevel - 1;
            q.max_level = nPredictedLevel;
            q.valid = 1;
        }
        std::vector<size_t> vNearIndices = F.GetFeaturesInArea(mTrackProjX[iMP], mTrackProjY[iMP], rr, nPredictedLevel - 1, nPredictedLevel);
        if (vNearIndices.empty()) continue;
        const uint8_t* MPdescriptor = mp_desc + iMP * 32;
        int bestDist = INT_MAX;
        int bestLevel = -1;
        int bestDist2 = INT_MAX;
        int bestLevel2 = -1;
        int bestIdx = -1;
        for (size_t idx : vNearIndices) {
            if (has[idx]) continue;
            const int dist = orbref_hamming(MPdescriptor, desc + idx * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestLevel2 = bestLevel;
                bestLevel = F.kp[idx].octave;
                bestIdx = (int)idx;
            } else if (dist < bestDist2) {
                bestLevel2 = F.kp[idx].octave;
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;
            has[bestIdx] = 1;
            match_of_feature[bestIdx] = (int32_t)iMP;
            nmatches++;
        }
    }
    return nmatches;
}

// Frame::isInFrustum(MapPoint *pMP, float viewingCosLimit), src/Frame.cc:138-199, for npoints map points of one
// frame with pose Tcw (mRcw, mtcw and mOw = -mRcw.t()*mtcw as Frame::UpdatePoseMatrices keeps them, :131-136).
// Outputs are the tracking fields the function leaves in the map point: mbTrackInView, mTrackProjX, mTrackProjY,
// mnTrackScaleLevel, mTrackViewCos (untouched slots stay 0).  Returns the number of points in view.
int orbref_is_in_frustum(const int32_t bounds[4], float fx, float fy, float cx, float cy, const float* Tcw /*4x4 row-major*/,
                         const float* mvScaleFactors, int mnScaleLevels, int npoints, const float* pos, const float* normal,
                         const float* minDistArr, const float* maxDistArr, float viewingCosLimit, uint8_t* mbTrackInView,
                         float* mTrackProjX, float* mTrackProjY, int32_t* mnTrackScaleLevel, float* mTrackViewCos) {
    const int mnMinX = bounds[0], mnMinY = bounds[1], mnMaxX = bounds[2], mnMaxY = bounds[3];
    float mRcw[9], mtcw[3], mOw[3];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) mRcw[3 * i + j] = Tcw[4 * i + j];
        mtcw[i] = Tcw[4 * i + 3];
    }
    centre(mRcw, mtcw, mOw);
    int nin = 0;
    for (int i = 0; i < npoints; i++) {
        mbTrackInView[i] = 0;
        mTrackProjX[i] = mTrackProjY[i] = mTrackViewCos[i] = 0.f;
        mnTrackScaleLevel[i] = 0;
        const float* P = pos + 3 * i;
        float Pc[3];
        rigid(Tcw, P, Pc);
        const float PcX = Pc[0];
        const float PcY = Pc[1];
        const float PcZ = Pc[2];
        if (PcZ < 0.0) continue;
        const float invz = 1.0 / PcZ;
        const float u = fx * PcX * invz + cx;
        const float v = fy * PcY * invz + cy;
        if (u < mnMinX || u > mnMaxX) continue;
        if (v < mnMinY || v > mnMaxY) continue;
        const float maxDistance = maxDistArr[i];
        const float minDistance = minDistArr[i];
        const float PO[3] = {P[0] - mOw[0], P[1] - mOw[1], P[2] - mOw[2]};
        const float dist = norm3(PO);
        if (dist < minDistance || dist > maxDistance) continue;
        float viewCos = dot3(PO, normal + 3 * i) / dist;
        if (viewCos < viewingCosLimit) continue;
        float ratio = dist / minDistance;
        const float* it = std::lower_bound(mvScaleFactors, mvScaleFactors + mnScaleLevels, ratio);
        int nPredictedLevel = (int)(it - mvScaleFactors);
        if (nPredictedLevel >= mnScaleLevels) nPredictedLevel = mnScaleLevels - 1;
        mbTrackInView[i] = 1;
        mTrackProjX[i] = u;
        mTrackProjY[i] = v;
        mnTrackScaleLevel[i] = nPredictedLevel;
        mTrackViewCos[i] = viewCos;
        nin++;
    }
    return nin;
}

// SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th),
// src/ORBmatcher.cc:297-418.  skip[i]: isBad() or in spAlreadyFound.  matched[idx] != 0: vpMatched[idx] set on entry.
int orbref_sbp_sim3(const void* kp, const uint8_t* desc, int n, const int32_t bounds[4], float fx, float fy, float cx, float cy,
                    const float* vfScaleFactors, int nlevels, const float* Scw /*4x4 row-major*/, const uint8_t* matched,
                    int npoints, const uint8_t* skip, const float* pos, const float* normal, const float* minDistArr,
                    const float* maxDistArr, const uint8_t* mp_desc, int th, int32_t* match_of_feature, void* probes_out) {
    FrameRef KF(kp, desc, n, bounds);
    ProbeRec* pr = (ProbeRec*)probes_out;
    const int nMaxLevel = nlevels - 1;
    // Decompose Scw
    float sR[9], Rcw[9], tcw[3], Ow[3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) sR[3 * i + j] = Scw[4 * i + j];
    const float scw = std::sqrt(dot3(sR, sR));
    const float inv = (float)(1. / scw);
    for (int i = 0; i < 9; i++) Rcw[i] = sR[i] * inv;
    for (int i = 0; i < 3; i++) tcw[i] = Scw[4 * i + 3] * inv;
    centre(Rcw, tcw, Ow);
    float T[16] = {Rcw[0], Rcw[1], Rcw[2], tcw[0], Rcw[3], Rcw[4], Rcw[5], tcw[1], Rcw[6], Rcw[7], Rcw[8], tcw[2], 0, 0, 0, 1};
    std::vector<char> has(n);
    for (int i = 0; i < n; i++) {
        has[i] = matched && matched[i];
        match_of_feature[i] = -1;
    }
    int nmatches = 0;
    for (int iMP = 0, iendMP = npoints; iMP < iendMP; iMP++) {
        clear_probe(pr ? pr + iMP : nullptr);
        if (skip[iMP]) continue;
        const float* p3Dw = pos + 3 * iMP;
        float p3Dc[3];
        rigid(T, p3Dw, p3Dc);
        if (p3Dc[2] < 0.0) continue;
        const float invz = 1 / p3Dc[2];
        const float x = p3Dc[0] * invz;
        const float y = p3Dc[1] * invz;
        const float u = fx * x + cx;
        const float v = fy * y + cy;
        if (!(u >= KF.mnMinX && u < KF.mnMaxX && v >= KF.mnMinY && v < KF.mnMaxY)) continue;  // KeyFrame::IsInImage
        const float maxDistance = maxDistArr[iMP];
        const float minDistance = minDistArr[iMP];
        float PO[3] = {p3Dw[0] - Ow[0], p3Dw[1] - Ow[1], p3Dw[2] - Ow[2]};
        const float dist3 = norm3(PO);
        if (dist3 < minDistance || dist3 > maxDistance) continue;
        if (dot3(PO, normal + 3 * iMP) < 0.5 * dist3) continue;
        const float ratio = dist3 / minDistance;
        const float* it = std::lower_bound(vfScaleFactors, vfScaleFactors + nlevels, ratio);
        const int nPredictedLevel = std::min(static_cast<int>(it - vfScaleFactors), nMaxLevel);
        const float radius = th * vfScaleFactors[nPredictedLevel];
        if (pr) {
            ProbeRec& q = pr[iMP];
            q.u = u;
            q.v = v;
            q.r = radius;
            q.min_level = nPredictedLevel - 1;
            q.max_level = nPredictedLevel;
            q.valid = 1;
        }
        std::vector<size_t> vIndices = KF.GetFeaturesInArea(u, v, radius, -1, -1);  // KeyFrame::GetFeaturesInArea, src/KeyFrame.cc:638-678
        if (vIndices.empty()) continue;
        const uint8_t* dMP = mp_desc + (size_t)iMP * 32;
        int bestDist = INT_MAX;
        int bestIdx = -1;
        for (size_t idx : vIndices) {
            if (has[idx]) continue;
            const int kpLevel = KF.kp[idx].octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            const int dist = orbref_hamming(dMP, desc + idx * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx = (int)idx;
            }
        }
        if (bestDist <= TH_LOW) {
            has[bestIdx] = 1;
            match_of_feature[bestIdx] = iMP;
            nmatches++;
        }
    }
    return nmatches;
}

// SearchByProjection(Frame &F1, Frame &F2, int windowSize, vector<MapPoint*> &vpMapPointMatches2), src/ORBmatcher.cc:530-605.
// skip1[i1]: F1.mvpMapPoints[i1] is NULL, bad, or already among F2's map points.  taken2: F2.mvpMapPoints set on entry.
int orbref_sbp_frames(const void* kp1, const uint8_t* desc1, int n1, const uint8_t* skip1, const float* pos1,
                      const float* Tc2w /*4x4 row-major*/, float fx, float fy, float cx, float cy, const void* kp2,
                      const uint8_t* desc2, int n2, const int32_t bounds[4], const uint8_t* taken2, int windowSize,
                      float mfNNratio, int32_t* match_of_feature2, void* probes_out) {
    FrameRef F1(kp1, desc1, n1, bounds), F2(kp2, desc2, n2, bounds);
    ProbeRec* pr = (ProbeRec*)probes_out;
    std::vector<char> has(n2);
    for (int i = 0; i < n2; i++) {
        has[i] = taken2 && taken2[i];
        match_of_feature2[i] = -1;
    }
    int nmatches = 0;
    for (size_t i1 = 0, iend1 = n1; i1 < iend1; i1++) {
        clear_probe(pr ? pr + i1 : nullptr);
        if (skip1[i1]) continue;
        int level1 = F1.kp[i1].octave;
        float x3Dc2[3];
        rigid(Tc2w, pos1 + 3 * i1, x3Dc2);
        const float xc2 = x3Dc2[0];
        const float yc2 = x3Dc2[1];
        const float invzc2 = 1.0 / x3Dc2[2];
        float u2 = fx * xc2 * invzc2 + cx;
        float v2 = fy * yc2 * invzc2 + cy;
        if (pr) {
            ProbeRec& q = pr[i1];
            q.u = u2;
            q.v = v2;
            q.r = windowSize;
            q.min_level = q.max_level = level1;
            q.valid = 1;
        }
        std::vector<size_t> vIndices2 = F2.GetFeaturesInArea(u2, v2, windowSize, level1, level1);
        if (vIndices2.empty()) continue;
        const uint8_t* d1 = desc1 + i1 * 32;
        int bestDist = INT_MAX;
        int bestDist2 = INT_MAX;
        int bestIdx2 = -1;
        for (size_t i2 : vIndices2) {
            if (has[i2]) continue;
            int dist = orbref_hamming(d1, desc2 + i2 * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestIdx2 = (int)i2;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        if (static_cast<float>(bestDist) <= static_cast<float>(bestDist2) * mfNNratio && bestDist <= TH_HIGH) {
            has[bestIdx2] = 1;
            match_of_feature2[bestIdx2] = (int32_t)i1;
            nmatches++;
        }
    }
    return nmatches;
}

namespace {

// Shared tail of the two tracking overloads: rotation histogram filter (:1608-1628, :1735-1754)
int rotation_filter(std::vector<int>* rotHist, int32_t* match_of_feature, int nmatches) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
        if (i != ind1 && i != ind2 && i != ind3) {
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                match_of_feature[rotHist[i][j]] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

}  // namespace

// SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, float th), src/ORBmatcher.cc:1518-1631.
// skip_last[i]: LastFrame.mvpMapPoints[i] is NULL or LastFrame.mvbOutlier[i].
int orbref_sbp_last_frame(const void* kp_cur, const uint8_t* desc_cur, int n_cur, const int32_t bounds[4], const uint8_t* taken_cur,
                          float fx, float fy, float cx, float cy, const float* Tcw /*4x4 row-major*/,
                          const float* mvScaleFactors, const void* kp_last, const uint8_t* desc_last, int n_last,
                          const uint8_t* skip_last, const float* pos_last, float th, int mbCheckOrientation,
                          int32_t* match_of_feature, void* probes_out) {
    FrameRef CurrentFrame(kp_cur, desc_cur, n_cur, bounds);
    const Kp* last = (const Kp*)kp_last;
    ProbeRec* pr = (ProbeRec*)probes_out;
    std::vector<char> has(n_cur);
    for (int i = 0; i < n_cur; i++) {
        has[i] = taken_cur && taken_cur[i];
        match_of_feature[i] = -1;
    }
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    for (size_t i = 0, iend = n_last; i < iend; i++) {
        clear_probe(pr ? pr + i : nullptr);
        if (skip_last[i]) continue;
        float x3Dc[3];
        rigid(Tcw, pos_last + 3 * i, x3Dc);
        const float xc = x3Dc[0];
        const float yc = x3Dc[1];
        const float invzc = 1.0 / x3Dc[2];
        float u = fx * xc * invzc + cx;
        float v = fy * yc * invzc + cy;
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
        if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        int nPredictedOctave = last[i].octave;
        float radius = th * mvScaleFactors[nPredictedOctave];
        if (pr) {
            ProbeRec& q = pr[i];
            q.u = u;
            q.v = v;
            q.r = radius;
            q.min_level = nPredictedOctave - 1;
            q.max_level = nPredictedOctave + 1;
            q.angle = last[i].angle;
            q.valid = 1;
        }
        std::vector<size_t> vIndices2 = CurrentFrame.GetFeaturesInArea(u, v, radius, nPredictedOctave - 1, nPredictedOctave + 1);
        if (vIndices2.empty()) continue;
        const uint8_t* dMP = desc_last + i * 32;
        int bestDist = INT_MAX;
        int bestIdx2 = -1;
        for (size_t i2 : vIndices2) {
            if (has[i2]) continue;
            int dist = orbref_hamming(dMP, desc_cur + i2 * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx2 = (int)i2;
            }
        }
        if (bestDist <= TH_HIGH) {
            has[bestIdx2] = 1;
            match_of_feature[bestIdx2] = (int32_t)i;
            nmatches++;
            if (mbCheckOrientation) {
                float rot = last[i].angle - CurrentFrame.kp[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (mbCheckOrientation) nmatches = rotation_filter(rotHist, match_of_feature, nmatches);
    return nmatches;
}

// SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, float th, int ORBdist),
// src/ORBmatcher.cc:1633-1757.  skip_kf[i]: pKF's map point i is NULL, bad or in sAlreadyFound.  kf_angle[i] =
// pKF->GetKeyPointUn(i).angle; mp_desc = pMP->GetDescriptor().
int orbref_sbp_keyframe(const void* kp_cur, const uint8_t* desc_cur, int n_cur, const int32_t bounds[4], const uint8_t* taken_cur,
                        float fx, float fy, float cx, float cy, const float* Tcw /*4x4 row-major*/, const float* mvScaleFactors,
                        int mnScaleLevels, int n_kf, const uint8_t* skip_kf, const float* pos, const float* minDistArr,
                        const uint8_t* mp_desc, const float* kf_angle, float th, int ORBdist, int mbCheckOrientation,
                        int32_t* match_of_feature, void* probes_out) {
    FrameRef CurrentFrame(kp_cur, desc_cur, n_cur, bounds);
    ProbeRec* pr = (ProbeRec*)probes_out;
    std::vector<char> has(n_cur);
    for (int i = 0; i < n_cur; i++) {
        has[i] = taken_cur && taken_cur[i];
        match_of_feature[i] = -1;
    }
    int nmatches = 0;
    float Rcw[9], tcw[3], Ow[3];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Rcw[3 * i + j] = Tcw[4 * i + j];
        tcw[i] = Tcw[4 * i + 3];
    }
    centre(Rcw, tcw, Ow);
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    for (size_t i = 0, iend = n_kf; i < iend; i++) {
        clear_probe(pr ? pr + i : nullptr);
        if (skip_kf[i]) continue;
        const float* x3Dw = pos + 3 * i;
        float x3Dc[3];
        rigid(Tcw, x3Dw, x3Dc);
        const float xc = x3Dc[0];
        const float yc = x3Dc[1];
        const float invzc = 1.0 / x3Dc[2];
        float u = fx * xc * invzc + cx;
        float v = fy * yc * invzc + cy;
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
        if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        float minDistance = minDistArr[i];
        float PO[3] = {x3Dw[0] - Ow[0], x3Dw[1] - Ow[1], x3Dw[2] - Ow[2]};
        float dist3D = norm3(PO);
        float ratio = dist3D / minDistance;
        const float* it = std::lower_bound(mvScaleFactors, mvScaleFactors + mnScaleLevels, ratio);
        const int nPredictedLevel = std::min(static_cast<int>(it - mvScaleFactors), mnScaleLevels - 1);
        float radius = th * mvScaleFactors[nPredictedLevel];
        if (pr) {
            ProbeRec& q = pr[i];
            q.u = u;
            q.v = v;
            q.r = radius;
            q.min_level = nPredictedLevel - 1;
            q.max_level = nPredictedLevel + 1;
            q.angle = kf_angle[i];
            q.valid = 1;
        }
        std::vector<size_t> vIndices2 = CurrentFrame.GetFeaturesInArea(u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1);
        if (vIndices2.empty()) continue;
        const uint8_t* dMP = mp_desc + i * 32;
        int bestDist = INT_MAX;
        int bestIdx2 = -1;
        for (size_t i2 : vIndices2) {
            if (has[i2]) continue;
            int dist = orbref_hamming(dMP, desc_cur + i2 * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx2 = (int)i2;
            }
        }
        if (bestDist <= ORBdist) {
            has[bestIdx2] = 1;
            match_of_feature[bestIdx2] = (int32_t)i;
            nmatches++;
            if (mbCheckOrientation) {
                float rot = kf_angle[i] - CurrentFrame.kp[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (mbCheckOrientation) nmatches = rotation_filter(rotHist, match_of_feature, nmatches);
    return nmatches;
}

// ---------------------------------------------------------------------------------------------
// ORBmatcher::Fuse (both forms) and ORBmatcher::SearchBySim3: per projected map point the most similar keypoint
// in the radius, with no interaction between map points.  What the reference then does with a match (Replace /
// AddObservation / AddMapPoint, src/ORBmatcher.cc:1127-1141, :1256-1270) touches only the map and stays with the
// caller; the functions below return bestIdx / bestDist per map point and the count the reference returns.
namespace {

// R (3x3 row-major) * x + t through cv::gemm's small-matrix path, as rigid() above
void rigid9(const float* R, const float* t, const float* x, float out[3]) {
    for (int i = 0; i < 3; i++) {
        const float* a = R + 3 * i;
        float t0 = a[0] * x[0] + a[1] * x[1] + a[2] * x[2];
        out[i] = (float)(t0 * 1.0 + t[i] * 1.0);
    }
}

// the loop body shared by :1054-1124, :1180-1253, :1340-1404 and :1423-1489 from "Depth must be positive" on:
// p3Dc = the point in the camera of KF, PO / normal = the viewing-angle test (normal == nullptr: SearchBySim3 has
// none), dist3D = the distance the scale prediction uses.  Returns false where the reference takes a `continue`
// before its scan; bestIdx stays -1 / bestDist INT_MAX when the scan finds nothing.
bool best_in_radius(FrameRef& KF, const uint8_t* desc, float fx, float fy, float cx, float cy, const float* vfScaleFactors,
                    int nlevels, const float p3Dc[3], bool double_invz, float dist3D, const float* PO, const float* normal,
                    float minDistance, float maxDistance, float th, const uint8_t* dMP, int& bestIdx, int& bestDist,
                    ProbeRec* q) {
    const int nMaxLevel = nlevels - 1;
    if (p3Dc[2] < 0.0f) return false;
    // :1061 "1/z" in float, :1190 / :1348 / :1431 "1.0/z" in double then stored to float: the same float (a
    // quotient of two floats rounded through double and then to float is the correctly rounded float quotient)
    const float invz = double_invz ? (float)(1.0 / p3Dc[2]) : 1 / p3Dc[2];
    const float x = p3Dc[0] * invz;
    const float y = p3Dc[1] * invz;
    const float u = fx * x + cx;
    const float v = fy * y + cy;
    if (!(u >= KF.mnMinX && u < KF.mnMaxX && v >= KF.mnMinY && v < KF.mnMaxY)) return false;  // KeyFrame::IsInImage, src/KeyFrame.cc:680-683
    if (dist3D < minDistance || dist3D > maxDistance) return false;
    if (normal && dot3(PO, normal) < 0.5 * dist3D) return false;
    const float ratio = dist3D / minDistance;
    const float* it = std::lower_bound(vfScaleFactors, vfScaleFactors + nlevels, ratio);
    const int nPredictedLevel = std::min(static_cast<int>(it - vfScaleFactors), nMaxLevel);
    const float radius = th * vfScaleFactors[nPredictedLevel];
    if (q) {
        q->u = u;
        q->v = v;
        q->r = radius;
        q->min_level = nPredictedLevel - 1;
        q->max_level = nPredictedLevel;
        q->valid = 1;
    }
    std::vector<size_t> vIndices = KF.GetFeaturesInArea(u, v, radius, -1, -1);  // KeyFrame::GetFeaturesInArea, src/KeyFrame.cc:638-678
    if (vIndices.empty()) return false;
    for (size_t idx : vIndices) {
        const int kpLevel = KF.kp[idx].octave;
        if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
        const int dist = orbref_hamming(dMP, desc + idx * 32);
        if (dist < bestDist) {
            bestDist = dist;
            bestIdx = (int)idx;
        }
    }
    return true;
}

}  // namespace

// Fuse(KeyFrame *pKF, vector<MapPoint *> &vpMapPoints, float th), src/ORBmatcher.cc:1027-1145.  skip[i]: the map point
// is NULL, isBad() or IsInKeyFrame(pKF).  Tcw: the keyframe's pose (GetRotation / GetTranslation; its camera centre
// Ow = -Rcw.t()*tcw, src/KeyFrame.cc:108).  Returns nFused; best_idx / best_dist per map point (-1 / INT_MAX where
// the scan did not run or found nothing).
int orbref_fuse_pose(const void* kp, const uint8_t* desc, int n, const int32_t bounds[4], float fx, float fy, float cx, float cy,
                     const float* vfScaleFactors, int nlevels, const float* Tcw /*4x4 row-major*/, int npoints,
                     const uint8_t* skip, const float* pos, const float* normal, const float* minDistArr,
                     const float* maxDistArr, const uint8_t* mp_desc, float th, int32_t* best_idx, int32_t* best_dist,
                     void* probes_out) {
    FrameRef KF(kp, desc, n, bounds);
    ProbeRec* pr = (ProbeRec*)probes_out;
    float Rcw[9], tcw[3], Ow[3];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Rcw[3 * i + j] = Tcw[4 * i + j];
        tcw[i] = Tcw[4 * i + 3];
    }
    centre(Rcw, tcw, Ow);
    int nFused = 0;
    for (int i = 0; i < npoints; i++) {
        clear_probe(pr ? pr + i : nullptr);
        best_idx[i] = -1;
        best_dist[i] = INT_MAX;
        if (skip[i]) continue;
        const float* p3Dw = pos + 3 * i;
        float p3Dc[3];
        rigid9(Rcw, tcw, p3Dw, p3Dc);
        float PO[3] = {p3Dw[0] - Ow[0], p3Dw[1] - Ow[1], p3Dw[2] - Ow[2]};
        const float dist3D = norm3(PO);
        int bestDist = INT_MAX, bestIdx = -1;
        if (!best_in_radius(KF, desc, fx, fy, cx, cy, vfScaleFactors, nlevels, p3Dc, false, dist3D, PO, normal + 3 * i,
                            minDistArr[i], maxDistArr[i], th, mp_desc + (size_t)i * 32, bestIdx, bestDist, pr ? pr + i : nullptr))
            continue;
        best_idx[i] = bestIdx;
        best_dist[i] = bestDist;
        if (bestDist <= TH_LOW) nFused++;
    }
    return nFused;
}

// Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint *> &vpPoints, float th), src/ORBmatcher.cc:1147-1276.
// skip[i]: isBad() or in spAlreadyFound.
int orbref_fuse_sim3(const void* kp, const uint8_t* desc, int n, const int32_t bounds[4], float fx, float fy, float cx, float cy,
                     const float* vfScaleFactors, int nlevels, const float* Scw /*4x4 row-major*/, int npoints,
                     const uint8_t* skip, const float* pos, const float* normal, const float* minDistArr,
                     const float* maxDistArr, const uint8_t* mp_desc, float th, int32_t* best_idx, int32_t* best_dist,
                     void* probes_out) {
    FrameRef KF(kp, desc, n, bounds);
    ProbeRec* pr = (ProbeRec*)probes_out;
    // Decompose Scw
    float sR[9], Rcw[9], tcw[3], Ow[3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) sR[3 * i + j] = Scw[4 * i + j];
    const float scw = std::sqrt(dot3(sR, sR));
    const float inv = (float)(1. / scw);
    for (int i = 0; i < 9; i++) Rcw[i] = sR[i] * inv;
    for (int i = 0; i < 3; i++) tcw[i] = Scw[4 * i + 3] * inv;
    centre(Rcw, tcw, Ow);
    int nFused = 0;
    for (int iMP = 0; iMP < npoints; iMP++) {
        clear_probe(pr ? pr + iMP : nullptr);
        best_idx[iMP] = -1;
        best_dist[iMP] = INT_MAX;
        if (skip[iMP]) continue;
        const float* p3Dw = pos + 3 * iMP;
        float p3Dc[3];
        rigid9(Rcw, tcw, p3Dw, p3Dc);
        float PO[3] = {p3Dw[0] - Ow[0], p3Dw[1] - Ow[1], p3Dw[2] - Ow[2]};
        const float dist3D = norm3(PO);
        int bestDist = INT_MAX, bestIdx = -1;
        if (!best_in_radius(KF, desc, fx, fy, cx, cy, vfScaleFactors, nlevels, p3Dc, true, dist3D, PO, normal + 3 * iMP,
                            minDistArr[iMP], maxDistArr[iMP], th, mp_desc + (size_t)iMP * 32, bestIdx, bestDist,
                            pr ? pr + iMP : nullptr))
            continue;
        best_idx[iMP] = bestIdx;
        best_dist[iMP] = bestDist;
        if (bestDist <= TH_LOW) nFused++;
    }
    return nFused;
}

// SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12, const float &s12, const cv::Mat &R12,
// const cv::Mat &t12, float th), src/ORBmatcher.cc:1278-1516.  skip1[i1]: pKF1's map point is NULL, bad or
// vbAlreadyMatched1[i1]; skip2[i2] likewise with vbAlreadyMatched2.  Both projections use pKF1's intrinsics
// (:1281-1284).  Outputs: vnMatch1 (n1), vnMatch2 (n2), matches12[i1] = idx2 for the pairs that agree (:1500-1513)
// else -1; returns nFound.
int orbref_search_by_sim3(const void* kp1, const uint8_t* desc1, int n1, const void* kp2, const uint8_t* desc2, int n2,
                          const int32_t bounds[4], float fx, float fy, float cx, float cy, const float* vfScaleFactors1,
                          int nlevels1, const float* vfScaleFactors2, int nlevels2, const float* T1w, const float* T2w,
                          float s12, const float* R12 /*9*/, const float* t12 /*3*/, float th, const uint8_t* skip1,
                          const float* pos1, const float* minDist1, const float* maxDist1, const uint8_t* mp_desc1,
                          const uint8_t* skip2, const float* pos2, const float* minDist2, const float* maxDist2,
                          const uint8_t* mp_desc2, int32_t* vnMatch1, int32_t* vnMatch2, int32_t* matches12, void* probes1_out,
                          void* probes2_out) {
    FrameRef KF1(kp1, desc1, n1, bounds), KF2(kp2, desc2, n2, bounds);
    ProbeRec* pr1 = (ProbeRec*)probes1_out;
    ProbeRec* pr2 = (ProbeRec*)probes2_out;
    float R1w[9], t1w[3], R2w[9], t2w[3];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            R1w[3 * i + j] = T1w[4 * i + j];
            R2w[3 * i + j] = T2w[4 * i + j];
        }
        t1w[i] = T1w[4 * i + 3];
        t2w[i] = T2w[4 * i + 3];
    }
    // sR12 = s12*R12, sR21 = (1.0/s12)*R12.t(): elementwise float products with (float)alpha; t21 = -sR21*t12: one
    // cv::gemm through the small-matrix path with alpha = -1
    float sR12[9], sR21[9], t21[3];
    const float a12 = (float)(double)s12, a21 = (float)(1.0 / s12);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            sR12[3 * i + j] = R12[3 * i + j] * a12;
            sR21[3 * i + j] = R12[3 * j + i] * a21;
        }
    for (int i = 0; i < 3; i++) {
        const float* a = sR21 + 3 * i;
        float t0 = a[0] * t12[0] + a[1] * t12[1] + a[2] * t12[2];
        t21[i] = (float)(t0 * -1.0);
    }
    for (int i = 0; i < n1; i++) vnMatch1[i] = -1;
    for (int i = 0; i < n2; i++) vnMatch2[i] = -1;
    // Transform from KF1 to KF2 and search
    for (int i1 = 0; i1 < n1; i1++) {
        clear_probe(pr1 ? pr1 + i1 : nullptr);
        if (skip1[i1]) continue;
        float p3Dc1[3], p3Dc2[3];
        rigid9(R1w, t1w, pos1 + 3 * i1, p3Dc1);
        rigid9(sR21, t21, p3Dc1, p3Dc2);
        const float dist3D = norm3(p3Dc2);
        int bestDist = INT_MAX, bestIdx = -1;
        if (!best_in_radius(KF2, desc2, fx, fy, cx, cy, vfScaleFactors2, nlevels2, p3Dc2, true, dist3D, nullptr, nullptr,
                            minDist1[i1], maxDist1[i1], th, mp_desc1 + (size_t)i1 * 32, bestIdx, bestDist, pr1 ? pr1 + i1 : nullptr))
            continue;
        if (bestDist <= TH_HIGH) vnMatch1[i1] = bestIdx;
    }
    // Transform from KF2 to KF1 and search
    for (int i2 = 0; i2 < n2; i2++) {
        clear_probe(pr2 ? pr2 + i2 : nullptr);
        if (skip2[i2]) continue;
        float p3Dc2[3], p3Dc1[3];
        rigid9(R2w, t2w, pos2 + 3 * i2, p3Dc2);
        rigid9(sR12, t12, p3Dc2, p3Dc1);
        const float dist3D = norm3(p3Dc1);
        int bestDist = INT_MAX, bestIdx = -1;
        if (!best_in_radius(KF1, desc1, fx, fy, cx, cy, vfScaleFactors1, nlevels1, p3Dc1, true, dist3D, nullptr, nullptr,
                            minDist2[i2], maxDist2[i2], th, mp_desc2 + (size_t)i2 * 32, bestIdx, bestDist, pr2 ? pr2 + i2 : nullptr))
            continue;
        if (bestDist <= TH_HIGH) vnMatch2[i2] = bestIdx;
    }
    // Check agreement
    int nFound = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        matches12[i1] = -1;
        int idx2 = vnMatch1[i1];
        if (idx2 >= 0) {
            int idx1 = vnMatch2[idx2];
            if (idx1 == i1) {
                matches12[i1] = idx2;
                nFound++;
            }
        }
    }
    return nFound;
}



// ---------------------------------------------------------------------------------------------
// cv::undistortPoints(src, dst, cameraMatrix, distCoeffs, R = Mat(), P = cameraMatrix) as OpenCV 2.4's
// cvUndistortPoints evaluates it (modules/imgproc/src/undistort.cpp; not part of the reference checkout,
// restated from its published algorithm): everything in double, camera matrix and coefficients converted
// from CV_32F, k[8] zero-filled past the coefficients given, iters = 5, RR = PP * identity by a 3x3 matrix
// product, result stored as float.  Called by Frame::UndistortKeyPoints (src/Frame.cc:290-320) and
// Frame::ComputeImageBounds (:322-350).
namespace {

void cv_undistort_points(const float* src_xy, float* dst_xy, int n, const float K[4] /*fx fy cx cy*/, const float* dist, int ndist) {
    double A[3][3] = {{K[0], 0, K[2]}, {0, K[1], K[3]}, {0, 0, 1}};
    double RR[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < ndist && i < 8; i++) k[i] = dist[i];
    const int iters = ndist > 0 ? 5 : 1;
    {
        double PP[3][3], T[3][3];
        memcpy(PP, A, sizeof(PP));
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double s0 = 0;
                for (int q = 0; q < 3; q++) s0 += PP[i][q] * RR[q][j];
                T[i][j] = s0;
            }
        memcpy(RR, T, sizeof(RR));
    }
    const double fx = A[0][0], fy = A[1][1], ifx = 1. / fx, ify = 1. / fy, cx = A[0][2], cy = A[1][2];
    for (int i = 0; i < n; i++) {
        double x, y, x0, y0;
        x = src_xy[2 * i];
        y = src_xy[2 * i + 1];
        x0 = x = (x - cx) * ifx;
        y0 = y = (y - cy) * ify;
        for (int j = 0; j < iters; j++) {
            double r2 = x * x + y * y;
            double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
            double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
        double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
        double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
        x = xx * ww;
        y = yy * ww;
        dst_xy[2 * i] = (float)x;
        dst_xy[2 * i + 1] = (float)y;
    }
}

}  // namespace

// Frame::UndistortKeyPoints: kp_un = kp with pt undistorted, or a plain copy when mDistCoef(0) == 0.
void orbref_undistort_keypoints(const void* kp, int n, const float K[4], const float dist[4], void* kp_un) {
    const Kp* in = (const Kp*)kp;
    Kp* out = (Kp*)kp_un;
    if (dist[0] == 0.0) {
        for (int i = 0; i < n; i++) out[i] = in[i];
        return;
    }
    std::vector<float> mat(2 * (size_t)n + 2);
    for (int i = 0; i < n; i++) {
        mat[2 * i] = in[i].x;
        mat[2 * i + 1] = in[i].y;
    }
    cv_undistort_points(mat.data(), mat.data(), n, K, dist, 4);
    for (int i = 0; i < n; i++) {
        Kp k = in[i];
        k.x = mat[2 * i];
        k.y = mat[2 * i + 1];
        out[i] = k;
    }
}

// Frame::ComputeImageBounds: bounds[4] = mnMinX, mnMinY, mnMaxX, mnMaxY
void orbref_image_bounds(const float K[4], const float dist[4], int cols, int rows, int32_t bounds[4]) {
    if (dist[0] != 0.0) {
        float mat[8] = {0.0f, 0.0f, (float)cols, 0.0f, 0.0f, (float)rows, (float)cols, (float)rows};
        cv_undistort_points(mat, mat, 4, K, dist, 4);
        bounds[0] = std::min(std::floor(mat[0]), std::floor(mat[4]));
        bounds[2] = std::max(std::ceil(mat[2]), std::ceil(mat[6]));
        bounds[1] = std::min(std::floor(mat[1]), std::floor(mat[3]));
        bounds[3] = std::max(std::ceil(mat[5]), std::ceil(mat[7]));
    } else {
        bounds[0] = 0;
        bounds[2] = cols;
        bounds[1] = 0;
        bounds[3] = rows;
    }
}

}  // extern "C"
