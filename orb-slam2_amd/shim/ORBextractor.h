// Drop-in replacement of the reference's include/ORBextractor.h: the same public class
// (constructor defaults, operator(), GetLevels, GetScaleFactor, score-type enum) so that Frame,
// Tracking, LocalMapping and LoopClosing compile unchanged, forwarding to the C ABI of
// include/orbhip.h.  Build with OpenCV 2.4 headers, or with -DORBHIP_SHIM_CV_STUB where OpenCV
// is absent (tests).
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <vector>
#ifdef ORBHIP_SHIM_CV_STUB
#include "cv_stub.h"
#else
#include <opencv/cv.h>
#endif

struct orbhip_extractor;

namespace ORB_SLAM {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures = 1000, float scaleFactor = 1.2f, int nlevels = 8, int scoreType = FAST_SCORE,
                 int fastTh = 20);
    ~ORBextractor();

    // Compute the ORB features and descriptors on an image (mask is accepted and, as in the
    // reference, has no effect on the result).
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints,
                    cv::OutputArray descriptors);

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return scaleFactor; }

    // GPU ordinal used by extractors constructed afterwards (default: environment variable
    // ORBHIP_DEVICE, else 0).
    static void SetDevice(int device);

protected:
    int nfeatures;
    double scaleFactor;
    int nlevels;
    int scoreType;
    int fastTh;

    orbhip_extractor* handle;  // created at the first call, when the image size is known
    int device;
    std::vector<unsigned char> descbuf;  // descriptor staging, sized at the first call

private:
    ORBextractor(const ORBextractor&);
    ORBextractor& operator=(const ORBextractor&);
};

}  // namespace ORB_SLAM

#endif
