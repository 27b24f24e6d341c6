// Forwarding implementation of ORB_SLAM::ORBextractor over the C ABI (include/orbhip.h).
// Behaviour at the boundary follows src/ORBextractor.cc:718-779 of the reference: empty image ->
// return with outputs untouched; image must be CV_8UC1 (assert); keypoints cleared and refilled;
// descriptors created N x 32 CV_8U, or released when N == 0.
#include "ORBextractor.h"

#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

#include "../../include/orbhip.h"

namespace ORB_SLAM {

static int g_device = -1;

void ORBextractor::SetDevice(int d) { g_device = d; }

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _scoreType, int _fastTh)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), scoreType(_scoreType), fastTh(_fastTh),
      handle(NULL) {
    device = g_device;
    if (device < 0) {
        const char* env = getenv("ORBHIP_DEVICE");
        device = env ? atoi(env) : 0;
    }
}

ORBextractor::~ORBextractor() {
    if (handle) orbhip_extractor_destroy(handle);
}

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray, std::vector<cv::KeyPoint>& _keypoints,
                              cv::OutputArray _descriptors) {
    if (_image.empty()) return;
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    if (!handle) {
        int rc = orbhip_extractor_create(&handle, nfeatures, (float)scaleFactor, nlevels, scoreType, fastTh, device,
                                         image.cols, image.rows, 1);
        if (rc != ORBHIP_OK) throw std::runtime_error(std::string("orbhip_extractor_create: ") + orbhip_last_error());
    }
    static_assert(sizeof(cv::KeyPoint) == sizeof(OrbKeypoint), "cv::KeyPoint must be the 28-byte 2.4 layout");
    // Keypoints land directly in the caller's vector (cv::KeyPoint and OrbKeypoint are the same 28 bytes);
    // descriptors go through a member buffer that is sized once.
    if (descbuf.size() < (size_t)nfeatures * 32) descbuf.resize((size_t)nfeatures * 32);
    _keypoints.resize(nfeatures);
    int n = 0;
    int rc = orbhip_extract(handle, image.ptr<unsigned char>(), image.rows, image.cols, (ptrdiff_t)image.step,
                            (OrbKeypoint*)(void*)&_keypoints[0], &descbuf[0], nfeatures, &n);
    if (rc != ORBHIP_OK) {
        _keypoints.clear();
        throw std::runtime_error(std::string("orbhip_extract: ") + orbhip_last_error());
    }
    _keypoints.resize(n);
    if (n == 0)
        _descriptors.release();
    else {
        _descriptors.create(n, 32, CV_8U);
        cv::Mat d = _descriptors.getMat();
        if (d.isContinuous())
            memcpy(d.ptr<unsigned char>(0), &descbuf[0], (size_t)n * 32);
        else
            for (int i = 0; i < n; i++) memcpy(d.ptr<unsigned char>(i), &descbuf[(size_t)i * 32], 32);
    }
}

}  // namespace ORB_SLAM
