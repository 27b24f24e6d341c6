/* orb_ref — CPU restatement of the ORB front end of the reference (TEST INFRASTRUCTURE ONLY).
 *
 * This library is the checker that the HIP path is compared against.  It is never linked,
 * loaded or called by the product library (liborbhip.so); only tests/, the smoke check and the
 * cpu_baseline leg of bench.py use it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path and
 * cannot be compiled here (it needs OpenCV 2.4 + ROS headers and libraries that are absent), so
 * this restatement is pinned only by its own known-answer tests, not by reference outputs.
 * It follows, line by line:
 *   src/ORBextractor.cc:73-77,124-151,154-194,457-511,513-520,522-707,709-716,718-779,781-822
 *   src/ORBmatcher.cc:42-45,461-487,1759-1800,1805-1821
 * and, for the third-party arithmetic those lines call, the OpenCV 2.4 scalar (non-SIMD)
 * semantics of cv::resize(INTER_LINEAR), cv::copyMakeBorder(BORDER_REFLECT_101), cv::FAST
 * (FAST-9/16 with non-max suppression), cv::KeyPointsFilter::retainBest (std::nth_element of
 * this toolchain's libstdc++), cv::GaussianBlur(7x7, sigma 2, 8-bit fixed point),
 * cv::fastAtan2 (2.4 polynomial), cvRound, and glibc cosf/sinf of the host.
 */
#ifndef ORB_REF_H
#define ORB_REF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Field-for-field cv::KeyPoint of OpenCV 2.4 (pt.x, pt.y, size, angle, response, octave, class_id). */
typedef struct OrbRefKeypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} OrbRefKeypoint;

typedef struct orbref_extractor orbref_extractor;

/* ORBextractor::ORBextractor (src/ORBextractor.cc:457-511). score_type: 0 = HARRIS, 1 = FAST. */
orbref_extractor* orbref_create(int nfeatures, float scale_factor, int nlevels, int score_type, int fast_th);
void orbref_destroy(orbref_extractor* e);

int orbref_levels(const orbref_extractor* e);         /* include/ORBextractor.h:47 */
float orbref_scale_factor(const orbref_extractor* e); /* include/ORBextractor.h:50 */

/* Tables built by the constructor. which: 0 = mnFeaturesPerLevel (int32[nlevels]),
 * 1 = mvScaleFactor (float[nlevels]), 2 = mvInvScaleFactor (float[nlevels]), 3 = umax (int32[16]),
 * 4 = integer Gaussian taps (int32[7]).  Returns the element count. */
int orbref_table(const orbref_extractor* e, int which, void* out);

/* ORBextractor::operator() (src/ORBextractor.cc:718-779).  Returns 0, or -1 if cap is too small
 * or the grid geometry is one the reference itself cannot run (zero cells, cells past the border).
 * An empty image (rows or cols 0) returns 0 and leaves the outputs untouched, like :721-722. */
int orbref_extract(orbref_extractor* e, const uint8_t* img, int rows, int cols, ptrdiff_t stride,
                   OrbRefKeypoint* kp_out, uint8_t* desc_out, int cap, int* n_out);

/* Stage taps, valid after orbref_extract.  Planes are (w+32) x (h+32) bytes, tightly packed. */
int orbref_level_size(const orbref_extractor* e, int level, int* w, int* h);
int orbref_plane(const orbref_extractor* e, int level, int blurred, uint8_t* out);
/* Per-level keypoints after retainBest, before orientation/scaling: x, y (level coords), response
 * (the FAST score as an integer; with HARRIS_SCORE the IEEE bits of the float response). */
int orbref_level_keypoints(const orbref_extractor* e, int level, int32_t* xyr, int cap);
/* Grid of the level: out[0]=levelCols, [1]=levelRows, [2]=cellW, [3]=cellH, [4]=nfeaturesCell. */
int orbref_level_grid(const orbref_extractor* e, int level, int32_t* out);
/* Per-cell raster-ordered candidates of the level as found by FAST (before retainBest):
 * counts[nCells] and, concatenated cell-major, triples x, y (cell image coords), response. */
int orbref_level_candidates(const orbref_extractor* e, int level, int32_t* counts, int32_t* xyr, int cap);
/* nToRetain per cell after the redistribution loop (src/ORBextractor.cc:644-670). */
int orbref_level_retain(const orbref_extractor* e, int level, int32_t* n_to_retain);

/* Primitives exposed for known-answer tests. */
int orbref_cvround(double v);
float orbref_fast_atan2(float y, float x);
void orbref_resize_linear(const uint8_t* src, int sw, int sh, ptrdiff_t sstride,
                          uint8_t* dst, int dw, int dh, ptrdiff_t dstride);
int orbref_reflect101(int p, int n);
/* FAST-9/16 + NMS over a whole image; returns count, writes x,y,score triples (raster order). */
int orbref_fast(const uint8_t* img, int rows, int cols, ptrdiff_t stride, int threshold, int32_t* xyr, int cap);
/* Closed-form corner measure: max over the 16 arcs of 9 of min(v - I_k) and of min(I_k - v). */
int orbref_fast_measure(const uint8_t* center, ptrdiff_t stride);
/* retainBest on a response array; perm_out receives the surviving original indices in order. */
int orbref_retain_best(const float* responses, int n, int n_points, int32_t* perm_out);
/* Observable result used by the extractor: first n_points elements after std::nth_element. */
int orbref_nth_prefix(const float* responses, int n, int n_points, int32_t* perm_out);
void orbref_gauss7(const uint8_t* padded_in, int w, int h, uint8_t* padded_out);
/* Study switch of the Gaussian column pass: 0 = scalar (the oracle's profile, the default), 1 = the fp32 /
 * round-to-even evaluation of an SSE2 build of OpenCV 2.4.  Process-wide; no test or bench sets it. */
void orbref_set_gauss_profile(int profile);
float orbref_ic_angle(const uint8_t* center, ptrdiff_t stride);
void orbref_descriptor(const uint8_t* center, ptrdiff_t stride, float angle_deg, uint8_t* desc32);
float orbref_cosf(float x);
float orbref_sinf(float x);

/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1805-1821), SWAR form. */
int orbref_hamming(const void* a, const void* b);
/* Best / second-best ascending scan (src/ORBmatcher.cc:461-485) over the full train set.
 * idx = -1, d1 = d2 = INT_MAX when nt == 0; d2 = INT_MAX when nt == 1. */
void orbref_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* d1, int32_t* d2);
/* Same scan over an index list (the form every Search* method uses). */
void orbref_knn2_indexed(const uint8_t* q, const uint8_t* t, const int32_t* cand, int ncand,
                         int32_t* idx, int32_t* d1, int32_t* d2);
/* Acceptance predicates (src/ORBmatcher.cc:237,487,119): form 0 = LT, 1 = LE, 2 = GT-reject. */
int orbref_ratio_accept(int d1, int d2, float ratio, int form);
/* ComputeThreeMaxima (src/ORBmatcher.cc:1759-1800) on bin sizes. */
void orbref_three_maxima(const int32_t* sizes, int L, int32_t* ind3);
/* Rotation-histogram bin of a match (src/ORBmatcher.cc:493-498). */
int orbref_rot_bin(float angle1, float angle2);

/* ---- Frame feature grid and the windowed matchers built on it (oracle/frame_ref.cpp) ----------
 * kp: 28-byte keypoint records {x,y,size,angle,response,octave,class_id}; bounds = {mnMinX, mnMinY,
 * mnMaxX, mnMaxY} (src/Frame.cc:322-350; {0,0,cols,rows} without distortion). */
/* Frame grid (src/Frame.cc:110-124,268-278) as CSR: cell = ix*48 + iy, 64*48+1 starts, n items max. */
void orbref_frame_grid(const void* kp, int n, const int32_t bounds[4], int32_t* cell_start, int32_t* items);
/* Frame::GetFeaturesInArea (src/Frame.cc:201-266); returns the count, writes at most cap indices. */
int orbref_features_in_area(const void* kp, int n, const int32_t bounds[4], float x, float y, float r, int min_level,
                            int max_level, int32_t* out, int cap);
/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:609-724); prev_matched_xy (2*n1 floats) is
 * read and updated like vbPrevMatched; returns nmatches, fills vnMatches12[n1]. */
int orbref_search_for_initialization(const void* kp1, const uint8_t* desc1, int n1, const void* kp2, const uint8_t* desc2,
                                     int n2, const int32_t bounds[4], float* prev_matched_xy, int window_size,
                                     float nnratio, int check_orientation, int32_t* matches12);
/* ORBmatcher::WindowSearch (src/ORBmatcher.cc:420-527); valid1[i] != 0 <=> F1.mvpMapPoints[i] is a good
 * map point; fills matches21[n2] with the F1 index assigned to each F2 feature (-1 = none). */
int orbref_window_search(const void* kp1, const uint8_t* desc1, int n1, const uint8_t* valid1, const void* kp2,
                         const uint8_t* desc2, int n2, const int32_t bounds[4], int window_size, int min_level,
                         int max_level, float nnratio, int check_orientation, int32_t* matches21);
/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:221-250): index of the descriptor with the
 * least median distance to the others (first one on ties), its median in *best_median. */
int orbref_distinctive_descriptor(const uint8_t* desc, int n, int32_t* best_median);

/* ---- Bag-of-words transform of the vocabulary tree (oracle/bow_ref.cpp) -------------------------
 * Tree given as nnodes nodes in id order (node 0 = root): parent[i], descriptor (32 B), weight
 * (used at leaves).  Children attach in id order; word ids count the leaves in id order.
 * weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY; scoring: 0 L1_NORM .. 5 DOT_PRODUCT (BowVector.h:36-53). */
typedef struct orbref_vocabulary orbref_vocabulary;
orbref_vocabulary* orbref_vocabulary_create(int nnodes, const int32_t* parent, const uint8_t* desc, const double* weight,
                                            int L, int weighting, int scoring);
void orbref_vocabulary_destroy(orbref_vocabulary* v);
/* transform(feature, word_id, weight, &nid, levelsup) (TemplatedVocabulary.h:1218-1259). */
void orbref_bow_transform_feature(const orbref_vocabulary* v, const uint8_t* feature, int levelsup, int32_t* word,
                                  int32_t* node, double* weight);
/* transform(features, BowVector, FeatureVector, levelsup) (:1125-1194): the BowVector as sorted
 * (word, value) pairs, the FeatureVector as sorted nodes with CSR lists of feature indices. */
int orbref_bow_transform(const orbref_vocabulary* v, const uint8_t* features, int n, int levelsup, int32_t* bow_words,
                         double* bow_values, int32_t* n_words, int32_t* fv_nodes, int32_t* fv_start, int32_t* fv_features,
                         int32_t* n_nodes);
/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (src/ORBmatcher.cc:158-295) and
 * SearchByBoW(KeyFrame*, KeyFrame*, ...) (:726-861).  FeatureVectors as sorted nodes with CSR lists of
 * feature indices; valid*: the keyframe's map point exists and is not bad. */
int orbref_search_by_bow_kf_frame(const void* kpKF, const uint8_t* descKF, int nKF, const uint8_t* validKF,
                                  const int32_t* fvKF_nodes, const int32_t* fvKF_start, const int32_t* fvKF_feat,
                                  int fvKF_n, const void* kpF, const uint8_t* descF, int nF, const int32_t* fvF_nodes,
                                  const int32_t* fvF_start, const int32_t* fvF_feat, int fvF_n, float nnratio,
                                  int check_orientation, int32_t* matchesF);
int orbref_search_by_bow_kf_kf(const void* kp1, const uint8_t* desc1, int n1, const uint8_t* valid1,
                               const int32_t* fv1_nodes, const int32_t* fv1_start, const int32_t* fv1_feat, int fv1_n,
                               const void* kp2, const uint8_t* desc2, int n2, const uint8_t* valid2,
                               const int32_t* fv2_nodes, const int32_t* fv2_start, const int32_t* fv2_feat, int fv2_n,
                               float nnratio, int check_orientation, int32_t* matches12);
/* L1Scoring::score (ScoringObject.cpp:23-68). */
double orbref_bow_score_l1(const int32_t* w1, const double* v1, int n1, const int32_t* w2, const double* v2, int n2);

#ifdef __cplusplus
}
#endif
#endif
