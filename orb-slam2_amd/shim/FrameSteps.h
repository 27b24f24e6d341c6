// Forwarding bodies of the three per-frame / per-map-point steps next to the matchers, for a host that holds the
// reference's objects: what replaces the loops of MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:221-250),
// Frame::UndistortKeyPoints (src/Frame.cc:290-320) and Frame::ComputeBoW (src/Frame.cc:281-288) when their data goes
// through liborbhip.so.  They share the calling thread's matcher handle (ORBmatcher::Accel).
#ifndef ORBHIP_FRAME_STEPS_H
#define ORBHIP_FRAME_STEPS_H
#include <vector>

#include "ORBmatcher.h"

struct orbhip_vocabulary;

namespace ORB_SLAM {

// Index into vDescriptors of the descriptor with the least median distance to the others (the first on ties), or -1
// for an empty list: the map point then keeps mDescriptor = vDescriptors[index].clone() as src/MapPoint.cc:252-255.
int DistinctiveDescriptorIndex(const std::vector<cv::Mat>& vDescriptors);

// F.mvKeysUn from F.mvKeys, F.mK and F.mDistCoef.
void UndistortKeyPoints(Frame& F);

// F.mBowVec / F.mFeatVec from F.mDescriptors and a vocabulary uploaded with orbhip_vocabulary_create (levelsup 4,
// L1 normalisation, as Frame::ComputeBoW asks of the ORB vocabulary); nothing if F.mBowVec is already filled.
void ComputeBoW(Frame& F, const orbhip_vocabulary* voc);

}  // namespace ORB_SLAM
#endif
