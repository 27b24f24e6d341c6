// Test program of the drop-in classes (built with -DORBHIP_SHIM_CV_STUB): reads a raw 8-bit image,
// runs ORB_SLAM::ORBextractor::operator() exactly as src/Frame.cc:61 does and writes keypoints and
// descriptors as raw binary for the Python test to compare with the oracle.
//   shim_test <in.raw> <rows> <cols> <nfeatures> <out.bin>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <ctime>
#include <string>
#include <vector>

#include "ORBextractor.h"
#include "ORBmatcher_accel.h"

// Stand-ins with the members of ORB_SLAM::Frame / MapPoint the forwarding matchers touch.
struct FakeMapPoint {
    bool bad;
    // tracking fields written by Frame::isInFrustum (include/MapPoint.h)
    bool mbTrackInView;
    int mnTrackScaleLevel;
    float mTrackViewCos, mTrackProjX, mTrackProjY;
    cv::Mat descriptor;
    FakeMapPoint() : bad(false), mbTrackInView(false), mnTrackScaleLevel(0), mTrackViewCos(1.f), mTrackProjX(0.f), mTrackProjY(0.f) {}
    bool isBad() const { return bad; }
    cv::Mat GetDescriptor() const { return descriptor; }
};
struct FakeFrame {
    std::vector<cv::KeyPoint> mvKeysUn;
    cv::Mat mDescriptors;
    std::vector<FakeMapPoint*> mvpMapPoints;
    std::vector<float> mvScaleFactors;
    static int mnMinX, mnMinY, mnMaxX, mnMaxY;
};
int FakeFrame::mnMinX, FakeFrame::mnMinY, FakeFrame::mnMaxX, FakeFrame::mnMaxY;

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

int main(int argc, char** argv) {
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
    int dist01 = n >= 2 ? ORB_SLAM::ORBmatcherAccel::DescriptorDistance(descriptors.row(0), descriptors.row(1)) : -1;
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
    FakeFrame F1, F2;
    FakeFrame::mnMinX = 0; FakeFrame::mnMinY = 0; FakeFrame::mnMaxX = cols; FakeFrame::mnMaxY = rows;
    F1.mvKeysUn = keys;
    F1.mDescriptors = descriptors;
    extractor(im2, cv::Mat(), F2.mvKeysUn, F2.mDescriptors);
    static FakeMapPoint good, bad;
    bad.bad = true;
    F1.mvpMapPoints.resize(F1.mvKeysUn.size());
    for (size_t i = 0; i < F1.mvpMapPoints.size(); i++) F1.mvpMapPoints[i] = i % 7 == 0 ? (FakeMapPoint*)0 : (i % 11 == 0 ? &bad : &good);
    F2.mvpMapPoints.assign(F2.mvKeysUn.size(), (FakeMapPoint*)0);
    ORB_SLAM::ORBmatcherAccel matcher(0.9f, true);
    std::vector<cv::Point2f> prev(F1.mvKeysUn.size());
    for (size_t i = 0; i < prev.size(); i++) prev[i] = F1.mvKeysUn[i].pt;
    std::vector<int> m12;
    int nInit = matcher.SearchForInitialization(F1, F2, prev, m12, 100);
    std::vector<FakeMapPoint*> mp2;
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
    std::vector<FakeMapPoint> pts(n);
    std::vector<FakeMapPoint*> vp(n);
    for (int i = 0; i < n; i++) {
        pts[i].bad = i % 13 == 0;
        pts[i].mbTrackInView = i % 5 != 0;
        pts[i].mnTrackScaleLevel = keys[i].octave;
        pts[i].mTrackViewCos = i % 3 == 0 ? 0.9985f : 0.99f;
        pts[i].mTrackProjX = keys[i].pt.x + 5.f;
        pts[i].mTrackProjY = keys[i].pt.y + 3.f;
        pts[i].descriptor = descriptors.row(i);
        vp[i] = &pts[i];
    }
    F2.mvScaleFactors.resize(8);
    F2.mvScaleFactors[0] = 1.f;
    for (int l = 1; l < 8; l++) F2.mvScaleFactors[l] = F2.mvScaleFactors[l - 1] * 1.2f;
    for (int i = 0; i < n2; i++) F2.mvpMapPoints[i] = i % 9 == 0 ? &good : (FakeMapPoint*)0;
    ORB_SLAM::ORBmatcherAccel matcher8(0.8f, true);
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
