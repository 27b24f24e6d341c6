#include "FrameSteps.h"

#include <cstring>
#include <stdexcept>

#include "../../include/orbhip.h"

namespace ORB_SLAM {

namespace {
void check(int rc) {
    if (rc != ORBHIP_OK) throw std::runtime_error(orbhip_last_error());
}
}  // namespace

int DistinctiveDescriptorIndex(const std::vector<cv::Mat>& vDescriptors) {
    const int N = (int)vDescriptors.size();
    if (N == 0) return -1;
    std::vector<unsigned char> rows((size_t)N * 32);
    for (int i = 0; i < N; i++) memcpy(&rows[(size_t)i * 32], vDescriptors[i].ptr<unsigned char>(0), 32);
    const int32_t start[2] = {0, N};
    int32_t best = -1, median = 0;
    check(orbhip_distinctive_descriptors(ORBmatcher::Accel(), rows.data(), start, 1, &best, &median));
    return best;
}

void UndistortKeyPoints(Frame& F) {
    const OrbCamera cam = {F.mK.at<float>(0, 0),      F.mK.at<float>(1, 1),      F.mK.at<float>(0, 2),      F.mK.at<float>(1, 2),
                           F.mDistCoef.at<float>(0), F.mDistCoef.at<float>(1), F.mDistCoef.at<float>(2), F.mDistCoef.at<float>(3)};
    F.mvKeysUn.resize(F.mvKeys.size());
    if (F.mvKeys.empty()) return;
    check(orbhip_undistort_keypoints(ORBmatcher::Accel(), (const OrbKeypoint*)&F.mvKeys[0], (int)F.mvKeys.size(), cam,
                                     (OrbKeypoint*)&F.mvKeysUn[0]));
}

void ComputeBoW(Frame& F, const orbhip_vocabulary* voc) {
    if (!F.mBowVec.empty()) return;
    const int n = F.mDescriptors.rows;
    std::vector<unsigned char> rows((size_t)(n > 0 ? n : 1) * 32);
    for (int i = 0; i < n; i++) memcpy(&rows[(size_t)i * 32], F.mDescriptors.ptr<unsigned char>(i), 32);
    std::vector<int32_t> word(n > 0 ? n : 1), node(n > 0 ? n : 1);
    std::vector<double> weight(n > 0 ? n : 1);
    check(orbhip_bow_transform(ORBmatcher::Accel(), voc, rows.data(), n, 4, word.data(), node.data(), weight.data()));
    F.mFeatVec.clear();
    for (int i = 0; i < n; i++) {
        if (!(weight[i] > 0)) continue;  // a word that carries no weight is left out of both containers
        F.mBowVec.addWeight(word[i], weight[i]);
        F.mFeatVec.addFeature(node[i], i);
    }
    F.mBowVec.normalize(DBoW2::L1);
}

}  // namespace ORB_SLAM
