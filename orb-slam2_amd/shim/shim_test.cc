// Test program of the drop-in classes (built with -DORBHIP_SHIM_CV_STUB): reads a raw 8-bit image,
// runs ORB_SLAM::ORBextractor::operator() exactly as src/Frame.cc:61 does and writes keypoints and
// descriptors as raw binary for the Python test to compare with the oracle.
//   shim_test <in.raw> <rows> <cols> <nfeatures> <out.bin>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ORBextractor.h"
#include "ORBmatcher_accel.h"

int main(int argc, char** argv) {
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
    fclose(f);
    return 0;
}
