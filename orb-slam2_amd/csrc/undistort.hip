// Keypoint undistortion and the undistorted image bounds of a frame, for gfx950:
//   Frame::UndistortKeyPoints  src/Frame.cc:290-320
//   Frame::ComputeImageBounds  src/Frame.cc:322-350
// Both call cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK) with the 3x3 CV_32F camera
// matrix and the 4x1 CV_32F distortion vector (k1, k2, p1, p2).  OpenCV 2.4 evaluates that call in
// double: the pixel is normalised with the reciprocal focal lengths, the distortion is removed by five
// fixed-point iterations, and the result is mapped back through P * R = mK * I.  undistort_point()
// below is that arithmetic, operation by operation (compiled without contraction), shared by the
// device kernel and the four-corner host computation of the bounds.
#include <hip/hip_runtime.h>

#include <cmath>

#include "matcher_internal.h"

namespace {

struct KpDev {
    float x, y, size, angle, response;
    int32_t octave, class_id;
};

__host__ __device__ inline void undistort_point(const OrbCamera& c, float px, float py, float& ox, float& oy) {
    const double fx = c.fx, fy = c.fy, cx = c.cx, cy = c.cy;
    const double k0 = c.k1, k1 = c.k2, k2 = c.p1, k3 = c.p2;  // k[0..3]; k[4..7] = 0 for a 4-vector
    const double k4 = 0, k5 = 0, k6 = 0, k7 = 0;
    const double ifx = 1. / fx, ify = 1. / fy;
    double x = px, y = py;
    const double x0 = x = (x - cx) * ifx;
    const double y0 = y = (y - cy) * ify;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((k7 * r2 + k6) * r2 + k5) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x);
        const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    // RR = P * I: rows (fx, 0, cx), (0, fy, cy), (0, 0, 1)
    const double xx = fx * x + 0.0 * y + cx;
    const double yy = 0.0 * x + fy * y + cy;
    const double ww = 1. / (0.0 * x + 0.0 * y + 1.0);
    ox = (float)(xx * ww);
    oy = (float)(yy * ww);
}

__global__ __launch_bounds__(256) void k_undistort(const KpDev* __restrict__ in, const int32_t* __restrict__ n_arr, int cap,
                                                   OrbCamera cam, KpDev* __restrict__ out) {
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= min(max(n_arr[f], 0), cap)) return;
    KpDev k = in[(long long)f * cap + i];
    if (cam.k1 != 0.0f) {  // "if(mDistCoef.at<float>(0)==0.0) mvKeysUn=mvKeys", :292-296
        float ux, uy;
        undistort_point(cam, k.x, k.y, ux, uy);
        k.x = ux;
        k.y = uy;
    }
    out[(long long)f * cap + i] = k;
}

}  // namespace

extern "C" {

int orbhip_undistort_keypoints_device(orbhip_matcher* m, const OrbKeypoint* d_kp, const int32_t* d_n, int batch, int cap,
                                      OrbCamera cam, OrbKeypoint* d_kp_un) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (batch < 0 || cap < 1) return orb_fail(ORBHIP_EINVAL, "bad batch / capacity");
    if (batch == 0) return ORBHIP_OK;
    if (!d_kp || !d_n || !d_kp_un) return orb_fail(ORBHIP_EINVAL, "null buffer");
    if (!(cam.fx != 0.0f) || !(cam.fy != 0.0f)) return orb_fail(ORBHIP_EINVAL, "zero focal length");
    HIPCHK(hipSetDevice(m->device));
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_undistort, dim3((cap + 255) / 256, batch), dim3(256), 0, m->stream, (const KpDev*)d_kp, d_n, cap, cam,
                       (KpDev*)d_kp_un);
    HIPCHK(hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_image_bounds(OrbCamera cam, int cols, int rows, OrbFrameBounds* out) {
    if (!out) return orb_fail(ORBHIP_EINVAL, "null output");
    if (cam.k1 != 0.0f) {
        if (!(cam.fx != 0.0f) || !(cam.fy != 0.0f)) return orb_fail(ORBHIP_EINVAL, "zero focal length");
        const float corners[4][2] = {{0.0f, 0.0f}, {(float)cols, 0.0f}, {0.0f, (float)rows}, {(float)cols, (float)rows}};
        float u[4][2];
        for (int i = 0; i < 4; i++) undistort_point(cam, corners[i][0], corners[i][1], u[i][0], u[i][1]);
        out->min_x = (int32_t)fminf(floorf(u[0][0]), floorf(u[2][0]));  // :337-340
        out->max_x = (int32_t)fmaxf(ceilf(u[1][0]), ceilf(u[3][0]));
        out->min_y = (int32_t)fminf(floorf(u[0][1]), floorf(u[1][1]));
        out->max_y = (int32_t)fmaxf(ceilf(u[2][1]), ceilf(u[3][1]));
    } else {
        out->min_x = 0;
        out->max_x = cols;
        out->min_y = 0;
        out->max_y = rows;
    }
    return ORBHIP_OK;
}

}  // extern "C"
