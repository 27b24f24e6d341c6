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

}  // extern "C"
