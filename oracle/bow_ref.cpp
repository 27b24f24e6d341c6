// CPU ORACLE (test infrastructure only; never linked into or loaded by the product library).
// PARITY UNPINNED: the reference ships no tests or fixtures for this path, cannot be built in this
// image, and its vocabulary file is not part of the checkout; this restatement is pinned only by its
// own known-answer tests on synthetic vocabulary trees.
//
// Literal restatement of the bag-of-words transform of the reference's vocabulary
// (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h), as Frame::ComputeBoW / KeyFrame::ComputeBoW call
// it (src/Frame.cc:281-288, src/KeyFrame.cc:82-91: transform(vCurrentDesc, mBowVec, mFeatVec, 4)):
//   Node                               TemplatedVocabulary.h:296-330
//   transform(feature, id, w, nid, levelsup)   :1218-1259
//   transform(features, v, fv, levelsup)       :1125-1194
//   FORB::distance                     Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101
//   BowVector::addWeight/addIfNotExist/normalize   BowVector.cpp:34-84
//   FeatureVector::addFeature          FeatureVector.cpp:31-45
//   L1Scoring::score                   ScoringObject.cpp:23-68
// and of the two matchers restricted to features under the same vocabulary node:
//   ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...)      src/ORBmatcher.cc:158-295
//   ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, ...)   src/ORBmatcher.cc:726-861
#include "orb_ref.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <map>
#include <utility>
#include <vector>

namespace {

typedef unsigned int WordId;
typedef double WordValue;
typedef unsigned int NodeId;
enum LNorm { L1, L2 };
enum WeightingType { TF_IDF, TF, IDF, BINARY };

struct Node {
    NodeId id;
    WordValue weight;
    std::vector<NodeId> children;
    NodeId parent;
    const uint8_t* descriptor;
    WordId word_id;
    Node() : id(0), weight(0), parent(0), descriptor(nullptr), word_id(0) {}
    bool isLeaf() const { return children.empty(); }
};

struct BowVector : std::map<WordId, WordValue> {
    void addWeight(WordId id, WordValue v) {
        iterator vit = this->lower_bound(id);
        if (vit != this->end() && !(this->key_comp()(id, vit->first))) vit->second += v;
        else this->insert(vit, value_type(id, v));
    }
    void addIfNotExist(WordId id, WordValue v) {
        iterator vit = this->lower_bound(id);
        if (vit == this->end() || (this->key_comp()(id, vit->first))) this->insert(vit, value_type(id, v));
    }
    void normalize(LNorm norm_type) {
        double norm = 0.0;
        if (norm_type == L1) {
            for (iterator it = begin(); it != end(); ++it) norm += fabs(it->second);
        } else {
            for (iterator it = begin(); it != end(); ++it) norm += it->second * it->second;
            norm = sqrt(norm);
        }
        if (norm > 0.0)
            for (iterator it = begin(); it != end(); ++it) it->second /= norm;
    }
};

struct FeatureVector : std::map<NodeId, std::vector<unsigned int> > {
    void addFeature(NodeId id, unsigned int i_feature) {
        iterator vit = this->lower_bound(id);
        if (vit != this->end() && vit->first == id) vit->second.push_back(i_feature);
        else {
            vit = this->insert(vit, value_type(id, std::vector<unsigned int>()));
            vit->second.push_back(i_feature);
        }
    }
};

}  // namespace

struct orbref_vocabulary {
    std::vector<Node> m_nodes;
    int m_L;
    WeightingType m_weighting;
    bool must;   // m_scoring_object->mustNormalize(norm)
    LNorm norm;

    // TemplatedVocabulary.h:1218-1259
    void transform(const uint8_t* feature, WordId& word_id, WordValue& weight, NodeId* nid, int levelsup) const {
        std::vector<NodeId> nodes;
        const int nid_level = m_L - levelsup;
        if (nid_level <= 0 && nid != NULL) *nid = 0;  // root
        NodeId final_id = 0;                          // root
        int current_level = 0;
        do {
            ++current_level;
            nodes = m_nodes[final_id].children;
            final_id = nodes[0];
            double best_d = orbref_hamming(feature, m_nodes[final_id].descriptor);
            for (std::vector<NodeId>::const_iterator nit = nodes.begin() + 1; nit != nodes.end(); ++nit) {
                NodeId id = *nit;
                double d = orbref_hamming(feature, m_nodes[id].descriptor);
                if (d < best_d) {
                    best_d = d;
                    final_id = id;
                }
            }
            if (nid != NULL && current_level == nid_level) *nid = final_id;
        } while (!m_nodes[final_id].isLeaf());
        word_id = m_nodes[final_id].word_id;
        weight = m_nodes[final_id].weight;
    }

    // TemplatedVocabulary.h:1125-1194
    void transform(const uint8_t* features, int n, BowVector& v, FeatureVector& fv, int levelsup) const {
        v.clear();
        fv.clear();
        if (m_nodes.size() <= 1) return;  // empty()
        if (m_weighting == TF || m_weighting == TF_IDF) {
            for (unsigned int i_feature = 0; i_feature < (unsigned)n; ++i_feature) {
                WordId id;
                NodeId nid;
                WordValue w;
                transform(features + (size_t)i_feature * 32, id, w, &nid, levelsup);
                if (w > 0) {
                    v.addWeight(id, w);
                    fv.addFeature(nid, i_feature);
                }
            }
            if (!v.empty() && !must) {
                const double nd = v.size();
                for (BowVector::iterator vit = v.begin(); vit != v.end(); vit++) vit->second /= nd;
            }
        } else {
            for (unsigned int i_feature = 0; i_feature < (unsigned)n; ++i_feature) {
                WordId id;
                NodeId nid;
                WordValue w;
                transform(features + (size_t)i_feature * 32, id, w, &nid, levelsup);
                if (w > 0) {
                    v.addIfNotExist(id, w);
                    fv.addFeature(nid, i_feature);
                }
            }
        }
        if (must) v.normalize(norm);
    }
};

extern "C" {

orbref_vocabulary* orbref_vocabulary_create(int nnodes, const int32_t* parent, const uint8_t* desc, const double* weight,
                                            int L, int weighting, int scoring) {
    if (nnodes < 1 || weighting < 0 || weighting > 3 || scoring < 0 || scoring > 5) return nullptr;
    orbref_vocabulary* v = new orbref_vocabulary;
    v->m_L = L;
    v->m_weighting = (WeightingType)weighting;
    // ScoringObject.h:72-87: every scoring normalises with L1 except L2_NORM (L2) and DOT_PRODUCT (none)
    v->must = scoring != 5;
    v->norm = scoring == 1 ? L2 : L1;
    v->m_nodes.resize(nnodes);
    // nodes arrive in id order and attach to their parent in that order (loadFromTextFile :1380-1415)
    for (int nid = 0; nid < nnodes; nid++) {
        Node& nd = v->m_nodes[nid];
        nd.id = nid;
        nd.descriptor = desc + (size_t)nid * 32;
        nd.weight = weight[nid];
        if (nid > 0) {
            nd.parent = parent[nid];
            v->m_nodes[parent[nid]].children.push_back(nid);
        }
    }
    WordId wid = 0;
    for (int nid = 1; nid < nnodes; nid++)
        if (v->m_nodes[nid].isLeaf()) v->m_nodes[nid].word_id = wid++;
    return v;
}

void orbref_vocabulary_destroy(orbref_vocabulary* v) { delete v; }

void orbref_bow_transform_feature(const orbref_vocabulary* v, const uint8_t* feature, int levelsup, int32_t* word,
                                  int32_t* node, double* weight) {
    WordId id;
    NodeId nid;
    WordValue w;
    v->transform(feature, id, w, &nid, levelsup);
    *word = (int32_t)id;
    *node = (int32_t)nid;
    *weight = w;
}

int orbref_bow_transform(const orbref_vocabulary* v, const uint8_t* features, int n, int levelsup, int32_t* bow_words,
                         double* bow_values, int32_t* n_words, int32_t* fv_nodes, int32_t* fv_start, int32_t* fv_features,
                         int32_t* n_nodes) {
    BowVector bv;
    FeatureVector fv;
    v->transform(features, n, bv, fv, levelsup);
    int i = 0;
    for (BowVector::const_iterator it = bv.begin(); it != bv.end(); ++it, ++i) {
        bow_words[i] = (int32_t)it->first;
        bow_values[i] = it->second;
    }
    *n_words = i;
    int k = 0, o = 0;
    for (FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it, ++k) {
        fv_nodes[k] = (int32_t)it->first;
        fv_start[k] = o;
        for (size_t j = 0; j < it->second.size(); j++) fv_features[o++] = (int32_t)it->second[j];
    }
    fv_start[k] = o;
    *n_nodes = k;
    return 0;
}

// L1Scoring::score (ScoringObject.cpp:23-68) on two sorted (word, value) lists.
double orbref_bow_score_l1(const int32_t* w1, const double* v1, int n1, const int32_t* w2, const double* v2, int n2) {
    int i = 0, j = 0;
    double score = 0;
    while (i < n1 && j < n2) {
        const double vi = v1[i], wi = v2[j];
        if (w1[i] == w2[j]) {
            score += fabs(vi - wi) - fabs(vi) - fabs(wi);
            ++i;
            ++j;
        } else if (w1[i] < w2[j]) {
            while (i < n1 && w1[i] < w2[j]) ++i;  // lower_bound
        } else {
            while (j < n2 && w2[j] < w1[i]) ++j;
        }
    }
    score = -score / 2.0;
    return score;
}

namespace {

struct KpB {
    float x, y, size, angle, response;
    int32_t octave, class_id;
};

FeatureVector fv_from_csr(const int32_t* nodes, const int32_t* start, const int32_t* feat, int nnodes) {
    FeatureVector fv;
    for (int j = 0; j < nnodes; j++)
        for (int i = start[j]; i < start[j + 1]; i++) fv.addFeature((NodeId)nodes[j], (unsigned)feat[i]);
    return fv;
}

const int BOW_HISTO_LENGTH = 30, BOW_TH_LOW = 20;

}  // namespace

// src/ORBmatcher.cc:158-295.  validKF[i] != 0 <=> vpMapPointsKF[i] exists and is not bad.
// matchesF[iF] = index in the keyframe whose map point goes to feature iF of the frame, or -1.
int orbref_search_by_bow_kf_frame(const void* kpKF_, const uint8_t* descKF, int nKF, const uint8_t* validKF,
                                  const int32_t* fvKF_nodes, const int32_t* fvKF_start, const int32_t* fvKF_feat,
                                  int fvKF_n, const void* kpF_, const uint8_t* descF, int nF, const int32_t* fvF_nodes,
                                  const int32_t* fvF_start, const int32_t* fvF_feat, int fvF_n, float mfNNratio,
                                  int mbCheckOrientation, int32_t* matchesF) {
    const KpB* kpKF = (const KpB*)kpKF_;
    const KpB* kpF = (const KpB*)kpF_;
    for (int i = 0; i < nF; i++) matchesF[i] = -1;
    FeatureVector vFeatVecKF = fv_from_csr(fvKF_nodes, fvKF_start, fvKF_feat, fvKF_n);
    FeatureVector FmFeatVec = fv_from_csr(fvF_nodes, fvF_start, fvF_feat, fvF_n);
    int nmatches = 0;
    std::vector<int> rotHist[BOW_HISTO_LENGTH];
    const float factor = 1.0f / BOW_HISTO_LENGTH;
    FeatureVector::iterator KFit = vFeatVecKF.begin(), Fit = FmFeatVec.begin();
    FeatureVector::iterator KFend = vFeatVecKF.end(), Fend = FmFeatVec.end();
    while (KFit != KFend && Fit != Fend) {
        if (KFit->first == Fit->first) {
            std::vector<unsigned int> vIndicesKF = KFit->second;
            std::vector<unsigned int> vIndicesF = Fit->second;
            for (size_t iKF = 0, iendKF = vIndicesKF.size(); iKF < iendKF; iKF++) {
                const unsigned int realIdxKF = vIndicesKF[iKF];
                if ((int)realIdxKF >= nKF) continue;  // vpMapPointsKF.at() throws -> continue
                if (!validKF[realIdxKF]) continue;
                const uint8_t* dKF = descKF + (size_t)realIdxKF * 32;
                int bestDist1 = INT_MAX;
                int bestIdxF = -1;
                int bestDist2 = INT_MAX;
                for (size_t iF = 0, iendF = vIndicesF.size(); iF < iendF; iF++) {
                    const unsigned int realIdxF = vIndicesF[iF];
                    if (matchesF[realIdxF] >= 0) continue;  // "if(vpMapPointMatches[realIdxF]) continue;"
                    const int dist = orbref_hamming(dKF, descF + (size_t)realIdxF * 32);
                    if (dist < bestDist1) {
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdxF = realIdxF;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                }
                if (bestDist1 <= BOW_TH_LOW) {
                    if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
                        matchesF[bestIdxF] = (int32_t)realIdxKF;
                        if (mbCheckOrientation) {
                            float rot = kpKF[realIdxKF].angle - kpF[bestIdxF].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = std::round(rot * factor);
                            if (bin == BOW_HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back(bestIdxF);
                        }
                        nmatches++;
                    }
                }
            }
            KFit++;
            Fit++;
        } else if (KFit->first < Fit->first) {
            KFit = vFeatVecKF.lower_bound(Fit->first);
        } else {
            Fit = FmFeatVec.lower_bound(KFit->first);
        }
    }
    if (mbCheckOrientation) {
        int32_t sizes[BOW_HISTO_LENGTH], ind[3];
        for (int i = 0; i < BOW_HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        orbref_three_maxima(sizes, BOW_HISTO_LENGTH, ind);
        for (int i = 0; i < BOW_HISTO_LENGTH; i++) {
            if (i == ind[0] || i == ind[1] || i == ind[2]) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                matchesF[rotHist[i][j]] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

// src/ORBmatcher.cc:726-861.  valid1 / valid2: the keyframes' map points exist and are not bad.
// matches12[idx1] = index in keyframe 2 whose map point goes to vpMatches12[idx1], or -1.
int orbref_search_by_bow_kf_kf(const void* kp1_, const uint8_t* desc1, int n1, const uint8_t* valid1,
                               const int32_t* fv1_nodes, const int32_t* fv1_start, const int32_t* fv1_feat, int fv1_n,
                               const void* kp2_, const uint8_t* desc2, int n2, const uint8_t* valid2,
                               const int32_t* fv2_nodes, const int32_t* fv2_start, const int32_t* fv2_feat, int fv2_n,
                               float mfNNratio, int mbCheckOrientation, int32_t* matches12) {
    const KpB* vKeysUn1 = (const KpB*)kp1_;
    const KpB* vKeysUn2 = (const KpB*)kp2_;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    std::vector<bool> vbMatched2(n2, false);
    FeatureVector vFeatVec1 = fv_from_csr(fv1_nodes, fv1_start, fv1_feat, fv1_n);
    FeatureVector vFeatVec2 = fv_from_csr(fv2_nodes, fv2_start, fv2_feat, fv2_n);
    std::vector<int> rotHist[BOW_HISTO_LENGTH];
    const float factor = 1.0f / BOW_HISTO_LENGTH;
    int nmatches = 0;
    FeatureVector::iterator f1it = vFeatVec1.begin(), f2it = vFeatVec2.begin();
    FeatureVector::iterator f1end = vFeatVec1.end(), f2end = vFeatVec2.end();
    while (f1it != f1end && f2it != f2end) {
        if (f1it->first == f2it->first) {
            for (size_t i1 = 0, iend1 = f1it->second.size(); i1 < iend1; i1++) {
                size_t idx1 = f1it->second[i1];
                if (!valid1[idx1]) continue;
                const uint8_t* d1 = desc1 + idx1 * 32;
                int bestDist1 = INT_MAX;
                int bestIdx2 = -1;
                int bestDist2 = INT_MAX;
                for (size_t i2 = 0, iend2 = f2it->second.size(); i2 < iend2; i2++) {
                    size_t idx2 = f2it->second[i2];
                    if (vbMatched2[idx2] || !valid2[idx2]) continue;
                    int dist = orbref_hamming(d1, desc2 + idx2 * 32);
                    if (dist < bestDist1) {
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdx2 = (int)idx2;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                }
                if (bestDist1 < BOW_TH_LOW) {
                    if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
                        matches12[idx1] = bestIdx2;
                        vbMatched2[bestIdx2] = true;
                        if (mbCheckOrientation) {
                            float rot = vKeysUn1[idx1].angle - vKeysUn2[bestIdx2].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = std::round(rot * factor);
                            if (bin == BOW_HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back((int)idx1);
                        }
                        nmatches++;
                    }
                }
            }
            f1it++;
            f2it++;
        } else if (f1it->first < f2it->first) {
            f1it = vFeatVec1.lower_bound(f2it->first);
        } else {
            f2it = vFeatVec2.lower_bound(f1it->first);
        }
    }
    if (mbCheckOrientation) {
        int32_t sizes[BOW_HISTO_LENGTH], ind[3];
        for (int i = 0; i < BOW_HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        orbref_three_maxima(sizes, BOW_HISTO_LENGTH, ind);
        for (int i = 0; i < BOW_HISTO_LENGTH; i++) {
            if (i == ind[0] || i == ind[1] || i == ind[2]) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                matches12[rotHist[i][j]] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

// ORBmatcher::CheckDistEpipolarLine, src/ORBmatcher.cc:139-156.  F12: 3x3 row-major (CV_32F); sigma2 =
// pKF2->GetSigma2(kp2.octave), a float (include/KeyFrame.h).
static bool CheckDistEpipolarLine(const KpB& kp1, const KpB& kp2, const float* F12, const float* mvLevelSigma2) {
    const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
    const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
    const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
    const float num = a * kp2.x + b * kp2.y + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * mvLevelSigma2[kp2.octave];
}

// ORBmatcher::SearchForTriangulation, src/ORBmatcher.cc:863-1025.  free1 / free2: the keyframe's feature holds no map
// point (pMP1 / pMP2 is NULL).  matches12[idx1] = vMatches12 after the orientation filter; vMatchedPairs are its
// entries >= 0 in index order (:1014-1022).  Returns nmatches.
int orbref_search_for_triangulation(const void* kp1_, const uint8_t* desc1, int n1, const uint8_t* free1,
                                    const int32_t* fv1_nodes, const int32_t* fv1_start, const int32_t* fv1_feat, int fv1_n,
                                    const void* kp2_, const uint8_t* desc2, int n2, const uint8_t* free2,
                                    const int32_t* fv2_nodes, const int32_t* fv2_start, const int32_t* fv2_feat, int fv2_n,
                                    const float* F12, const float* mvLevelSigma2, int mbCheckOrientation, int32_t* matches12) {
    const KpB* vKeysUn1 = (const KpB*)kp1_;
    const KpB* vKeysUn2 = (const KpB*)kp2_;
    FeatureVector vFeatVec1 = fv_from_csr(fv1_nodes, fv1_start, fv1_feat, fv1_n);
    FeatureVector vFeatVec2 = fv_from_csr(fv2_nodes, fv2_start, fv2_feat, fv2_n);
    int nmatches = 0;
    std::vector<bool> vbMatched2(n2, false);
    std::vector<int> vMatches12(n1, -1);
    std::vector<int> rotHist[BOW_HISTO_LENGTH];
    const float factor = 1.0f / BOW_HISTO_LENGTH;
    FeatureVector::iterator f1it = vFeatVec1.begin(), f2it = vFeatVec2.begin();
    FeatureVector::iterator f1end = vFeatVec1.end(), f2end = vFeatVec2.end();
    while (f1it != f1end && f2it != f2end) {
        if (f1it->first == f2it->first) {
            for (size_t i1 = 0, iend1 = f1it->second.size(); i1 < iend1; i1++) {
                size_t idx1 = f1it->second[i1];
                // If there is already a MapPoint skip
                if (!free1[idx1]) continue;
                const KpB& kp1 = vKeysUn1[idx1];
                const uint8_t* d1 = desc1 + idx1 * 32;
                std::vector<std::pair<int, size_t> > vDistIndex;
                for (size_t i2 = 0, iend2 = f2it->second.size(); i2 < iend2; i2++) {
                    size_t idx2 = f2it->second[i2];
                    // If we have already matched or there is a MapPoint skip
                    if (vbMatched2[idx2] || !free2[idx2]) continue;
                    const int dist = orbref_hamming(d1, desc2 + idx2 * 32);
                    if (dist > BOW_TH_LOW) continue;
                    vDistIndex.push_back(std::make_pair(dist, idx2));
                }
                if (vDistIndex.empty()) continue;
                std::sort(vDistIndex.begin(), vDistIndex.end());
                int BestDist = vDistIndex.front().first;
                int DistTh = std::round(2 * BestDist);
                for (size_t id = 0; id < vDistIndex.size(); id++) {
                    if (vDistIndex[id].first > DistTh) break;
                    int currentIdx2 = (int)vDistIndex[id].second;
                    const KpB& kp2 = vKeysUn2[currentIdx2];
                    if (CheckDistEpipolarLine(kp1, kp2, F12, mvLevelSigma2)) {
                        vbMatched2[currentIdx2] = true;
                        vMatches12[idx1] = currentIdx2;
                        nmatches++;
                        if (mbCheckOrientation) {
                            float rot = kp1.angle - kp2.angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = std::round(rot * factor);
                            if (bin == BOW_HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back((int)idx1);
                        }
                        break;
                    }
                }
            }
            f1it++;
            f2it++;
        } else if (f1it->first < f2it->first) {
            f1it = vFeatVec1.lower_bound(f2it->first);
        } else {
            f2it = vFeatVec2.lower_bound(f1it->first);
        }
    }
    if (mbCheckOrientation) {
        int32_t sizes[BOW_HISTO_LENGTH], ind[3];
        for (int i = 0; i < BOW_HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        orbref_three_maxima(sizes, BOW_HISTO_LENGTH, ind);
        for (int i = 0; i < BOW_HISTO_LENGTH; i++) {
            if (i == ind[0] || i == ind[1] || i == ind[2]) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                vMatches12[rotHist[i][j]] = -1;
                nmatches--;
            }
        }
    }
    for (int i = 0; i < n1; i++) matches12[i] = vMatches12[i];
    return nmatches;
}

}  // extern "C"
