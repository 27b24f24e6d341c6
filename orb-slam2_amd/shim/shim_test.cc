// Test program of the drop-in classes (built with -DORBHIP_SHIM_CV_STUB): reads a raw 8-bit image,
// runs ORB_SLAM::ORBextractor::operator() exactly as src/Frame.cc:61 does and writes keypoints and
// descriptors as raw binary for the Python test to compare with the oracle.
//   shim_test <in.raw> <rows> <cols> <nfeatures> <out.bin>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <map>
#include <set>
#include <cstdlib>
#include <ctime>
#include <string>
#include <vector>

#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "FrameSteps.h"
#include "KeyFrame.h"
#include "MapPoint.h"
#include "../../include/orbhip.h"

// Frame, KeyFrame and MapPoint are the stand-ins of shim/standin (the reference's names and accessors over plain
// members); the static members of Frame are defined here.
using ORB_SLAM::Frame;
using ORB_SLAM::KeyFrame;
using ORB_SLAM::MapPoint;
int Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY;
float Frame::fx, Frame::fy, Frame::cx, Frame::cy;

// shim_test --latency <in.raw> <rows> <cols> <nfeatures> <calls>: the call of src/Frame.cc:61 repeated on one
// image, one frame per call as the tracking thread makes it; prints the median, 10th and 90th percentile of the
// wall-clock time per call in microseconds and the keypoint count (after 20 warm-up calls).
static int latency_mode(int argc, char** argv) {
    if (argc < 7) return 2;
    const int rows = atoi(argv[3]), cols = atoi(argv[4]), nf = atoi(argv[5]), calls = atoi(argv[6]);
    std::vector<unsigned char> buf((size_t)rows * cols);
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(buf.data(), 1, buf.size(), f) != buf.size()) return 3;
    fclose(f);
    cv::Mat im(rows, cols, CV_8U, buf.data(), (size_t)cols);
    ORB_SLAM::ORBextractor extractor(nf, 1.2f, 8, ORB_SLAM::ORBextractor::FAST_SCORE, 20);
    std::vector<cv::KeyPoint> keys;
    cv::Mat descriptors;
    std::vector<double> us;
    for (int i = 0; i < calls + 20; i++) {
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        extractor(im, cv::Mat(), keys, descriptors);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (i >= 20) us.push_back((t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3);
    }
    std::sort(us.begin(), us.end());
    printf("{\"calls\": %d, \"median_us\": %.1f, \"p10_us\": %.1f, \"p90_us\": %.1f, \"keypoints\": %d}\n", calls,
           us[us.size() / 2], us[us.size() / 10], us[us.size() * 9 / 10], (int)keys.size());
    return 0;
}


// shim_test --matchers <mode> <in.bin> <out.bin>: one forwarding body of shim/ORBmatcher.cc on inputs a
// test wrote, results back for comparison with the oracle.  Both files are sequences of blobs, each an int32
// byte count followed by the bytes.
typedef std::vector<std::vector<unsigned char> > Blobs;
static bool read_blobs(const char* path, Blobs& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    int nb;
    while (fread(&nb, 4, 1, f) == 1) {
        std::vector<unsigned char> b(nb > 0 ? nb : 0);
        if (nb > 0 && fread(b.data(), 1, nb, f) != (size_t)nb) return false;
        out.push_back(b);
    }
    fclose(f);
    return true;
}
static void write_blob(FILE* f, const void* p, size_t bytes) {
    int nb = (int)bytes;
    fwrite(&nb, 4, 1, f);
    if (bytes) fwrite(p, 1, bytes, f);
}
template <class T> static const T* as(const std::vector<unsigned char>& b) { return (const T*)b.data(); }
static cv::Mat mat_f32(int rows, int cols, const float* src) {
    cv::Mat m(rows, cols, CV_32F);
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < cols; j++) m.at<float>(i, j) = src[i * cols + j];
    return m;
}
static cv::Mat mat_desc(const std::vector<unsigned char>& b) {
    const int n = (int)(b.size() / 32);
    cv::Mat m(n, 32, CV_8U);
    if (n) memcpy(m.data, b.data(), (size_t)n * 32);
    return m;
}
static std::vector<cv::KeyPoint> keypoints(const std::vector<unsigned char>& b) {
    std::vector<cv::KeyPoint> k(b.size() / sizeof(cv::KeyPoint));
    if (!k.empty()) memcpy(&k[0], b.data(), k.size() * sizeof(cv::KeyPoint));
    return k;
}
static DBoW2::FeatureVector feature_vector(const std::vector<unsigned char>& nodes, const std::vector<unsigned char>& start,
                                           const std::vector<unsigned char>& feat) {
    DBoW2::FeatureVector fv;
    const int cnt = (int)(nodes.size() / 4);
    for (int j = 0; j < cnt; j++)
        for (int k = as<int>(start)[j]; k < as<int>(start)[j + 1]; k++) fv.addFeature(as<int>(nodes)[j], as<int>(feat)[k]);
    return fv;
}

static int matchers_mode(int argc, char** argv) {
    if (argc < 5) return 2;
    const std::string mode = argv[2];
    Blobs in;
    if (!read_blobs(argv[3], in)) return 3;
    FILE* out = fopen(argv[4], "wb");
    if (!out) return 3;
    static MapPoint pre;  // the map point a feature holds on entry
    if (mode == "sim3" || mode == "frames" || mode == "last" || mode == "keyframe") {
        // blobs: 0 int params {th, window, orbdist, check_ori}, 1 float params {fx, fy, cx, cy, ratio, th}, 2 bounds,
        // 3 target keypoints, 4 target descriptors, 5 taken, 6 pose (4x4), 7 scale factors, 8 skip kind per point
        // (0 = usable, 1 = NULL, 2 = bad, 3 = already found / outlier), 9 positions, 10 normals, 11 min distance,
        // 12 max distance, 13 point descriptors, 14 source keypoints (one per point)
        const int* ip = as<int>(in[0]);
        const float* fp = as<float>(in[1]);
        const int* bd = as<int>(in[2]);
        Frame::mnMinX = bd[0]; Frame::mnMinY = bd[1]; Frame::mnMaxX = bd[2]; Frame::mnMaxY = bd[3];
        Frame::fx = fp[0]; Frame::fy = fp[1]; Frame::cx = fp[2]; Frame::cy = fp[3];
        ORB_SLAM::ORBmatcher matcher(fp[4], ip[3] != 0);
        Frame T;  // the target frame
        T.mvKeysUn = T.mvKeys = keypoints(in[3]);
        T.mDescriptors = mat_desc(in[4]);
        const int n = (int)T.mvKeysUn.size();
        T.mvpMapPoints.assign(n, (MapPoint*)0);
        for (int i = 0; i < n; i++)
            if (in[5][i]) T.mvpMapPoints[i] = &pre;
        T.mTcw = mat_f32(4, 4, as<float>(in[6]));
        const int nlev = (int)(in[7].size() / 4);
        T.mvScaleFactors.assign(as<float>(in[7]), as<float>(in[7]) + nlev);
        T.mnScaleLevels = nlev;
        const int npts = (int)in[8].size();
        std::vector<MapPoint> pts(npts);
        std::vector<MapPoint*> vp(npts);
        std::set<MapPoint*> found;
        cv::Mat pdesc = mat_desc(in[13]);
        for (int i = 0; i < npts; i++) {
            pts[i].mbBad = in[8][i] == 2;
            pts[i].mWorldPos = mat_f32(3, 1, as<float>(in[9]) + 3 * i);
            pts[i].mNormalVector = mat_f32(3, 1, as<float>(in[10]) + 3 * i);
            pts[i].mfMinDistance = as<float>(in[11])[i];
            pts[i].mfMaxDistance = as<float>(in[12])[i];
            pts[i].mDescriptor = pdesc.row(i);
            vp[i] = in[8][i] == 1 ? (MapPoint*)0 : &pts[i];
            if (in[8][i] == 3) found.insert(&pts[i]);
        }
        int nmatches = -1;
        std::vector<MapPoint*> result;
        if (mode == "sim3") {
            KeyFrame KF;
            KF.fx = fp[0]; KF.fy = fp[1]; KF.cx = fp[2]; KF.cy = fp[3];
            KF.mvKeysUn = T.mvKeysUn;
            KF.mDescriptors = T.mDescriptors;
            KF.mvScaleFactors = T.mvScaleFactors;
            // "already found" points of this overload sit in vpMatched on entry: give each one a feature that is taken
            result = T.mvpMapPoints;
            std::set<MapPoint*>::iterator it = found.begin();
            for (int i = 0; i < n && it != found.end(); i++)
                if (result[i] == &pre) result[i] = *it++;
            for (int i = 0; i < npts; i++)
                if (!vp[i]) vp[i] = &pre;  // this overload has no NULL test: a NULL point is modelled as already found
            nmatches = matcher.SearchByProjection(&KF, T.mTcw, vp, result, ip[0]);
        } else if (mode == "frames") {
            Frame F1;
            F1.mvKeysUn = keypoints(in[14]);
            F1.mDescriptors = pdesc;
            F1.mvpMapPoints = vp;
            // an "already found" point is one F2 already holds
            std::set<MapPoint*>::iterator it = found.begin();
            for (int i = 0; i < n && it != found.end(); i++)
                if (T.mvpMapPoints[i] == &pre) T.mvpMapPoints[i] = *it++;
            nmatches = matcher.SearchByProjection(F1, T, ip[1], result);
        } else if (mode == "last") {
            Frame L;
            L.mvKeys = L.mvKeysUn = keypoints(in[14]);
            L.mDescriptors = pdesc;
            L.mvpMapPoints = vp;
            L.mvbOutlier.assign(npts, false);
            for (int i = 0; i < npts; i++)
                if (in[8][i] >= 2) L.mvbOutlier[i] = true;  // kinds 2 and 3 stand for outliers here
            for (int i = 0; i < npts; i++) pts[i].mbBad = false;
            nmatches = matcher.SearchByProjection(T, (const Frame&)L, fp[5]);
            result = T.mvpMapPoints;
        } else {
            KeyFrame KF;
            KF.mvKeysUn = keypoints(in[14]);
            KF.mvpMapPoints = vp;
            nmatches = matcher.SearchByProjection(T, &KF, found, fp[5], ip[2]);
            result = T.mvpMapPoints;
        }
        std::vector<int> owner(n ? n : 1, -1);
        for (int i = 0; i < n; i++) {
            MapPoint* q = result[i];
            if (!q) continue;
            owner[i] = (q >= &pts[0] && q < &pts[0] + npts) ? (int)(q - &pts[0]) : -2;
            if (in[5][i]) owner[i] = -2;  // held on entry (by `pre` or by an already-found point)
        }
        write_blob(out, &nmatches, 4);
        write_blob(out, owner.data(), sizeof(int) * n);
    } else if (mode == "fuse_pose" || mode == "fuse_sim3") {
        // blobs 0-13 as above (1: {fx, fy, cx, cy, unused, th}; 6: Tcw or Scw; 8: 0 usable, 1 NULL, 2 bad, 3 already in
        // the keyframe), 14: per point the feature that holds it on entry (kind 3) or -1.  A taken feature holds a
        // map point of its own.  Map point ids in the output: candidate i = i, the point held by feature j = npts + j.
        const float* fp = as<float>(in[1]);
        const int* bd = as<int>(in[2]);
        Frame::mnMinX = bd[0]; Frame::mnMinY = bd[1]; Frame::mnMaxX = bd[2]; Frame::mnMaxY = bd[3];
        ORB_SLAM::ORBmatcher matcher;
        KeyFrame KF;
        KF.fx = fp[0]; KF.fy = fp[1]; KF.cx = fp[2]; KF.cy = fp[3];
        KF.mvKeysUn = keypoints(in[3]);
        KF.mDescriptors = mat_desc(in[4]);
        const int nlev = (int)(in[7].size() / 4);
        KF.mvScaleFactors.assign(as<float>(in[7]), as<float>(in[7]) + nlev);
        const int n = (int)KF.mvKeysUn.size(), npts = (int)in[8].size();
        std::vector<MapPoint> pts(npts + n);
        KF.mvpMapPoints.assign(n, (MapPoint*)0);
        for (int j = 0; j < n; j++)
            if (in[5][j]) {
                KF.mvpMapPoints[j] = &pts[npts + j];
                pts[npts + j].mObservations[&KF] = j;
            }
        cv::Mat pdesc = mat_desc(in[13]);
        std::vector<MapPoint*> vp(npts);
        for (int i = 0; i < npts; i++) {
            pts[i].mbBad = in[8][i] == 2;
            pts[i].mWorldPos = mat_f32(3, 1, as<float>(in[9]) + 3 * i);
            pts[i].mNormalVector = mat_f32(3, 1, as<float>(in[10]) + 3 * i);
            pts[i].mfMinDistance = as<float>(in[11])[i];
            pts[i].mfMaxDistance = as<float>(in[12])[i];
            pts[i].mDescriptor = pdesc.row(i);
            vp[i] = in[8][i] == 1 ? (MapPoint*)0 : &pts[i];
            const int host = as<int>(in[14])[i];
            if (in[8][i] == 3 && host >= 0) {
                pts[npts + host].mObservations.clear();
                KF.mvpMapPoints[host] = &pts[i];
                pts[i].mObservations[&KF] = host;
            }
        }
        cv::Mat T = mat_f32(4, 4, as<float>(in[6]));
        int nFused;
        if (mode == "fuse_pose") {
            KF.SetPose(T);
            nFused = matcher.Fuse(&KF, vp, fp[5]);
        } else {
            nFused = matcher.Fuse(&KF, T, (const std::vector<MapPoint*>&)vp, fp[5]);
        }
        std::vector<int> slot(n ? n : 1, -1), isbad(npts + n ? npts + n : 1, 0), obs(npts + n ? npts + n : 1, -1);
        for (int j = 0; j < n; j++)
            if (KF.mvpMapPoints[j]) slot[j] = (int)(KF.mvpMapPoints[j] - &pts[0]);
        for (int i = 0; i < npts + n; i++) {
            isbad[i] = pts[i].mbBad;
            obs[i] = pts[i].GetIndexInKeyFrame(&KF);
        }
        write_blob(out, &nFused, 4);
        write_blob(out, slot.data(), sizeof(int) * n);
        write_blob(out, isbad.data(), sizeof(int) * (npts + n));
        write_blob(out, obs.data(), sizeof(int) * (npts + n));
    } else if (mode == "sim3search") {
        // blobs: 0 float {fx, fy, cx, cy, s12, th}, 1 bounds, 2 scale factors, 3 T1w, 4 T2w, 5 R12, 6 t12, then per keyframe
        // (7.., 15..): keypoints, descriptors, kind per keypoint's map point (0 usable, 1 NULL, 2 bad, 3 (keyframe 1 only)
        // matched on entry), positions, min distance, max distance, point descriptors, partner (keyframe 1, kind 3: the
        // feature of keyframe 2 its match on entry is observed at, or -1)
        const float* fp = as<float>(in[0]);
        const int* bd = as<int>(in[1]);
        Frame::mnMinX = bd[0]; Frame::mnMinY = bd[1]; Frame::mnMaxX = bd[2]; Frame::mnMaxY = bd[3];
        ORB_SLAM::ORBmatcher matcher;
        KeyFrame KF[2];
        std::vector<MapPoint> pts[2];
        const int nlev = (int)(in[2].size() / 4);
        for (int s = 0; s < 2; s++) {
            const int o = 7 + 8 * s;
            KF[s].fx = fp[0]; KF[s].fy = fp[1]; KF[s].cx = fp[2]; KF[s].cy = fp[3];
            KF[s].mvKeysUn = keypoints(in[o]);
            KF[s].mDescriptors = mat_desc(in[o + 1]);
            KF[s].mvScaleFactors.assign(as<float>(in[2]), as<float>(in[2]) + nlev);
            KF[s].SetPose(mat_f32(4, 4, as<float>(in[3 + s])));
            const int n = (int)KF[s].mvKeysUn.size();
            pts[s].resize(n);
            KF[s].mvpMapPoints.assign(n, (MapPoint*)0);
            cv::Mat pdesc = mat_desc(in[o + 6]);
            for (int i = 0; i < n; i++) {
                pts[s][i].mbBad = in[o + 2][i] == 2;
                pts[s][i].mWorldPos = mat_f32(3, 1, as<float>(in[o + 3]) + 3 * i);
                pts[s][i].mfMinDistance = as<float>(in[o + 4])[i];
                pts[s][i].mfMaxDistance = as<float>(in[o + 5])[i];
                pts[s][i].mDescriptor = pdesc.row(i);
                if (in[o + 2][i] != 1) KF[s].mvpMapPoints[i] = &pts[s][i];
            }
        }
        const int n1 = (int)pts[0].size();
        std::vector<MapPoint> prior(n1 ? n1 : 1);  // the matches vpMatches12 holds on entry
        std::vector<MapPoint*> vpMatches12(n1, (MapPoint*)0);
        for (int i = 0; i < n1; i++)
            if (in[9][i] == 3) {
                vpMatches12[i] = &prior[i];
                const int partner = as<int>(in[14])[i];
                if (partner >= 0) prior[i].mObservations[&KF[1]] = partner;
            }
        cv::Mat R12 = mat_f32(3, 3, as<float>(in[5])), t12 = mat_f32(3, 1, as<float>(in[6]));
        const float s12 = fp[4];
        const int nFound = matcher.SearchBySim3(&KF[0], &KF[1], vpMatches12, s12, R12, t12, fp[5]);
        std::vector<int> idx(n1 ? n1 : 1, -1);
        for (int i = 0; i < n1; i++) {
            MapPoint* q = vpMatches12[i];
            if (!q) continue;
            idx[i] = (!pts[1].empty() && q >= &pts[1][0] && q < &pts[1][0] + pts[1].size()) ? (int)(q - &pts[1][0]) : -3;
        }
        write_blob(out, &nFound, 4);
        write_blob(out, idx.data(), sizeof(int) * n1);
    } else if (mode == "triangulate") {
        // blobs: 0 F12 (9 floats), 1 sigma2 per level, 2 int {check_ori}, then per keyframe (3.., 9..): keypoints,
        // descriptors, free (1 = the feature holds no map point), fv nodes, fv start, fv features
        ORB_SLAM::ORBmatcher matcher(0.6f, as<int>(in[2])[0] != 0);
        static MapPoint held;
        KeyFrame KF[2];
        for (int s = 0; s < 2; s++) {
            const int o = 3 + 6 * s;
            KF[s].mvKeysUn = keypoints(in[o]);
            KF[s].mDescriptors = mat_desc(in[o + 1]);
            const int n = (int)KF[s].mvKeysUn.size();
            KF[s].mvpMapPoints.assign(n, (MapPoint*)0);
            for (int i = 0; i < n; i++)
                if (!in[o + 2][i]) KF[s].mvpMapPoints[i] = &held;
            KF[s].mFeatVec = feature_vector(in[o + 3], in[o + 4], in[o + 5]);
            const int nlev = (int)(in[1].size() / 4);
            KF[s].mvLevelSigma2.assign(as<float>(in[1]), as<float>(in[1]) + nlev);
            KF[s].mvScaleFactors.assign(nlev, 1.f);
        }
        std::vector<cv::KeyPoint> k1, k2;
        std::vector<std::pair<size_t, size_t> > pairs;
        const int nmatches = matcher.SearchForTriangulation(&KF[0], &KF[1], mat_f32(3, 3, as<float>(in[0])), k1, k2, pairs);
        std::vector<int> flat(pairs.size() * 2 + 1);
        for (size_t i = 0; i < pairs.size(); i++) {
            flat[2 * i] = (int)pairs[i].first;
            flat[2 * i + 1] = (int)pairs[i].second;
        }
        write_blob(out, &nmatches, 4);
        write_blob(out, flat.data(), sizeof(int) * pairs.size() * 2);
        write_blob(out, k1.empty() ? 0 : &k1[0], k1.size() * sizeof(cv::KeyPoint));
        write_blob(out, k2.empty() ? 0 : &k2[0], k2.size() * sizeof(cv::KeyPoint));
    } else if (mode == "bow_kf_f" || mode == "bow_kf_kf") {
        // blobs: 0 float {ratio}, 1 int {check_ori}, then per side (2.., 8..): keypoints, descriptors, valid, fv nodes,
        // fv start, fv features
        ORB_SLAM::ORBmatcher matcher(as<float>(in[0])[0], as<int>(in[1])[0] != 0);
        KeyFrame KF[2];
        std::vector<MapPoint> pts[2];
        for (int s = 0; s < 2; s++) {
            const int o = 2 + 6 * s;
            KF[s].mvKeysUn = keypoints(in[o]);
            KF[s].mDescriptors = mat_desc(in[o + 1]);
            const int n = (int)KF[s].mvKeysUn.size();
            pts[s].resize(n);
            KF[s].mvpMapPoints.assign(n, (MapPoint*)0);
            for (int i = 0; i < n; i++) {
                const unsigned char v = in[o + 2].empty() ? 1 : in[o + 2][i];  // 1 valid, 0 NULL, 2 bad
                pts[s][i].mbBad = v == 2;
                if (v) KF[s].mvpMapPoints[i] = &pts[s][i];
            }
            KF[s].mFeatVec = feature_vector(in[o + 3], in[o + 4], in[o + 5]);
        }
        int nmatches;
        std::vector<MapPoint*> res;
        std::vector<int> idx;
        if (mode == "bow_kf_f") {
            Frame F;
            F.mvKeys = F.mvKeysUn = KF[1].mvKeysUn;
            F.mDescriptors = KF[1].mDescriptors;
            F.mvpMapPoints.assign(F.mvKeys.size(), (MapPoint*)0);
            F.mFeatVec = KF[1].mFeatVec;
            // a keyframe whose map-point vector is shorter than its features (the .at() guard of the reference)
            if (!KF[0].mvpMapPoints.empty()) KF[0].mvpMapPoints.resize(KF[0].mvpMapPoints.size() - KF[0].mvpMapPoints.size() / 16);
            nmatches = matcher.SearchByBoW(&KF[0], F, res);
            idx.assign(res.size() ? res.size() : 1, -1);
            for (size_t i = 0; i < res.size(); i++)
                if (res[i]) idx[i] = (int)(res[i] - &pts[0][0]);
        } else {
            nmatches = matcher.SearchByBoW(&KF[0], &KF[1], res);
            idx.assign(res.size() ? res.size() : 1, -1);
            for (size_t i = 0; i < res.size(); i++)
                if (res[i]) idx[i] = (int)(res[i] - &pts[1][0]);
        }
        write_blob(out, &nmatches, 4);
        write_blob(out, idx.data(), sizeof(int) * res.size());
    } else if (mode == "distinct") {
        ORB_SLAM::ORBmatcher matcher;
        cv::Mat d = mat_desc(in[0]);
        std::vector<cv::Mat> v;
        for (int i = 0; i < d.rows; i++) v.push_back(d.row(i));
        const int best = ORB_SLAM::DistinctiveDescriptorIndex(v);
        write_blob(out, &best, 4);
    } else if (mode == "undistort") {
        // blobs: 0 keypoints, 1 float {fx, fy, cx, cy, k1, k2, p1, p2}
        ORB_SLAM::ORBmatcher matcher;
        Frame F;
        F.mvKeys = keypoints(in[0]);
        const float* c = as<float>(in[1]);
        const float K[9] = {c[0], 0.f, c[2], 0.f, c[1], c[3], 0.f, 0.f, 1.f};
        F.mK = mat_f32(3, 3, K);
        F.mDistCoef = mat_f32(4, 1, c + 4);
        ORB_SLAM::UndistortKeyPoints(F);
        write_blob(out, F.mvKeysUn.empty() ? 0 : &F.mvKeysUn[0], F.mvKeysUn.size() * sizeof(cv::KeyPoint));
    } else if (mode == "computebow") {
        // blobs: 0 int {L}, 1 parent, 2 node descriptors, 3 weights (double), 4 frame descriptors
        ORB_SLAM::ORBmatcher matcher;
        orbhip_vocabulary* voc = 0;
        const int nn = (int)(in[1].size() / 4);
        if (orbhip_vocabulary_create(&voc, 0, nn, as<int>(in[1]), in[2].data(), as<double>(in[3]), as<int>(in[0])[0]) != ORBHIP_OK) {
            fprintf(stderr, "%s\n", orbhip_last_error());
            return 7;
        }
        Frame F;
        F.mDescriptors = mat_desc(in[4]);
        ORB_SLAM::ComputeBoW(F, voc);
        orbhip_vocabulary_destroy(voc);
        std::vector<int> words, nodes, start(1, 0), feat;
        std::vector<double> vals;
        for (DBoW2::BowVector::iterator it = F.mBowVec.begin(); it != F.mBowVec.end(); ++it) {
            words.push_back((int)it->first);
            vals.push_back(it->second);
        }
        for (DBoW2::FeatureVector::iterator it = F.mFeatVec.begin(); it != F.mFeatVec.end(); ++it) {
            nodes.push_back((int)it->first);
            for (size_t k = 0; k < it->second.size(); k++) feat.push_back((int)it->second[k]);
            start.push_back((int)feat.size());
        }
        write_blob(out, words.data(), words.size() * 4);
        write_blob(out, vals.data(), vals.size() * 8);
        write_blob(out, nodes.data(), nodes.size() * 4);
        write_blob(out, start.data(), start.size() * 4);
        write_blob(out, feat.data(), feat.size() * 4);
    } else {
        return 2;
    }
    fclose(out);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "--matchers") {
        try {
            return matchers_mode(argc, argv);
        } catch (const std::exception& e) {
            fprintf(stderr, "%s\n", e.what());
            return 9;
        }
    }
    if (argc > 1 && std::string(argv[1]) == "--latency") return latency_mode(argc, argv);
    if (argc < 6) return 2;
    const int rows = atoi(argv[2]), cols = atoi(argv[3]), nf = atoi(argv[4]);
    std::vector<unsigned char> buf((size_t)rows * cols);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(buf.data(), 1, buf.size(), f) != buf.size()) return 3;
    fclose(f);
    cv::Mat im(rows, cols, CV_8U, buf.data(), (size_t)cols);
    ORB_SLAM::ORBextractor extractor(nf, 1.2f, 8, ORB_SLAM::ORBextractor::FAST_SCORE, 20);
    std::vector<cv::KeyPoint> keys;
    cv::Mat descriptors;
    extractor(im, cv::Mat(), keys, descriptors);
    // an empty image leaves the outputs untouched
    std::vector<cv::KeyPoint> keys2(3);
    cv::Mat d2;
    extractor(cv::Mat(), cv::Mat(), keys2, d2);
    if (keys2.size() != 3) return 4;
    if (extractor.GetLevels() != 8 || extractor.GetScaleFactor() != 1.2f) return 5;
    int n = (int)keys.size();
    if (n != descriptors.rows) return 6;
    int dist01 = n >= 2 ? ORB_SLAM::ORBmatcher::DescriptorDistance(descriptors.row(0), descriptors.row(1)) : -1;
    f = fopen(argv[5], "wb");
    fwrite(&n, 4, 1, f);
    fwrite(&dist01, 4, 1, f);
    if (n) fwrite(&keys[0], sizeof(cv::KeyPoint), n, f);
    for (int i = 0; i < n; i++) fwrite(descriptors.ptr<unsigned char>(i), 1, 32, f);
    // the windowed matchers through their forwarding bodies: frame 2 is the image shifted by (5, 3)
    std::vector<unsigned char> buf2(buf.size());
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) buf2[(size_t)y * cols + x] = buf[(size_t)((y + rows - 3) % rows) * cols + (x + cols - 5) % cols];
    cv::Mat im2(rows, cols, CV_8U, buf2.data(), (size_t)cols);
    Frame F1, F2;
    Frame::mnMinX = 0; Frame::mnMinY = 0; Frame::mnMaxX = cols; Frame::mnMaxY = rows;
    F1.mvKeysUn = keys;
    F1.mDescriptors = descriptors;
    extractor(im2, cv::Mat(), F2.mvKeysUn, F2.mDescriptors);
    static MapPoint good, bad;
    bad.mbBad = true;
    F1.mvpMapPoints.resize(F1.mvKeysUn.size());
    for (size_t i = 0; i < F1.mvpMapPoints.size(); i++) F1.mvpMapPoints[i] = i % 7 == 0 ? (MapPoint*)0 : (i % 11 == 0 ? &bad : &good);
    F2.mvpMapPoints.assign(F2.mvKeysUn.size(), (MapPoint*)0);
    ORB_SLAM::ORBmatcher matcher(0.9f, true);
    std::vector<cv::Point2f> prev(F1.mvKeysUn.size());
    for (size_t i = 0; i < prev.size(); i++) prev[i] = F1.mvKeysUn[i].pt;
    std::vector<int> m12;
    int nInit = matcher.SearchForInitialization(F1, F2, prev, m12, 100);
    std::vector<MapPoint*> mp2;
    int nWin = matcher.WindowSearch(F1, F2, 50, mp2, 0);
    int n2 = (int)F2.mvKeysUn.size();
    fwrite(&n2, 4, 1, f);
    if (n2) fwrite(&F2.mvKeysUn[0], sizeof(cv::KeyPoint), n2, f);
    for (int i = 0; i < n2; i++) fwrite(F2.mDescriptors.ptr<unsigned char>(i), 1, 32, f);
    fwrite(&nInit, 4, 1, f);
    if (n) fwrite(&m12[0], 4, n, f);
    if (n) fwrite(&prev[0], 8, n, f);
    fwrite(&nWin, 4, 1, f);
    for (int i = 0; i < n2; i++) {
        int has = mp2[i] != 0;
        fwrite(&has, 4, 1, f);
    }
    // SearchByProjection(Frame&, vector<MapPoint*>&, th): one map point per keypoint of frame 1, predicted at
    // that keypoint moved by the known image shift, matched into frame 2
    std::vector<MapPoint> pts(n);
    std::vector<MapPoint*> vp(n);
    for (int i = 0; i < n; i++) {
        pts[i].mbBad = i % 13 == 0;
        pts[i].mbTrackInView = i % 5 != 0;
        pts[i].mnTrackScaleLevel = keys[i].octave;
        pts[i].mTrackViewCos = i % 3 == 0 ? 0.9985f : 0.99f;
        pts[i].mTrackProjX = keys[i].pt.x + 5.f;
        pts[i].mTrackProjY = keys[i].pt.y + 3.f;
        pts[i].mDescriptor = descriptors.row(i);
        vp[i] = &pts[i];
    }
    F2.mvScaleFactors.resize(8);
    F2.mvScaleFactors[0] = 1.f;
    for (int l = 1; l < 8; l++) F2.mvScaleFactors[l] = F2.mvScaleFactors[l - 1] * 1.2f;
    for (int i = 0; i < n2; i++) F2.mvpMapPoints[i] = i % 9 == 0 ? &good : (MapPoint*)0;
    ORB_SLAM::ORBmatcher matcher8(0.8f, true);
    int nProj = matcher8.SearchByProjection(F2, vp, 3.f);
    fwrite(&nProj, 4, 1, f);
    for (int i = 0; i < n2; i++) {
        int owner = -1;  // -2: the map point the feature held on entry
        if (F2.mvpMapPoints[i] == &good) owner = -2;
        else if (F2.mvpMapPoints[i]) owner = (int)(F2.mvpMapPoints[i] - &pts[0]);
        fwrite(&owner, 4, 1, f);
    }
    fclose(f);
    return 0;
}
