// FAST-9/16 corner measure shared by the extractor kernels.
#ifndef ORBHIP_FAST_MEASURE_H
#define ORBHIP_FAST_MEASURE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace orbk {

// FAST-9/16 corner measure M(p) = max(A, B): A = max over the 16 arcs of nine contiguous circle
// pixels of min(v - I_k), B the same for (I_k - v).  p is a corner at threshold t iff M > t and
// cv::FAST's score for it is M - 1 whatever t was (fast_score.cpp cornerScore<16>), so one
// measure plane per level serves both the fastTh pass and the threshold-7 retry of
// src/ORBextractor.cc:607-614.
__device__ __forceinline__ int fast_measure(const uint8_t* __restrict__ p, int s) {
    const int v = p[0];
    int d[16];
    d[0] = v - p[3 * s];
    d[1] = v - p[3 * s + 1];
    d[2] = v - p[2 * s + 2];
    d[3] = v - p[s + 3];
    d[4] = v - p[3];
    d[5] = v - p[-s + 3];
    d[6] = v - p[-2 * s + 2];
    d[7] = v - p[-3 * s + 1];
    d[8] = v - p[-3 * s];
    d[9] = v - p[-3 * s - 1];
    d[10] = v - p[-2 * s - 2];
    d[11] = v - p[-s - 3];
    d[12] = v - p[-3];
    d[13] = v - p[s - 3];
    d[14] = v - p[2 * s - 2];
    d[15] = v - p[3 * s - 1];
    int mn3[16], mx3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mn3[k] = min(d[k], min(d[(k + 1) & 15], d[(k + 2) & 15]));
        mx3[k] = max(d[k], max(d[(k + 1) & 15], d[(k + 2) & 15]));
    }
    int A = -256, Bn = 256;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        A = max(A, min(mn3[k], min(mn3[(k + 3) & 15], mn3[(k + 6) & 15])));
        Bn = min(Bn, max(mx3[k], max(mx3[(k + 3) & 15], mx3[(k + 6) & 15])));
    }
    return max(A, -Bn);
}

}  // namespace orbk
#endif
