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

__device__ __forceinline__ int min3_i32(int a, int b, int c) {
    int r;
    asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int max3_i32(int a, int b, int c) {
    int r;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// The same measure for a pixel of an LDS tile whose row pitch is a run-time value: seven row
// pointers (column -3 of rows -3..3) so that every circle pixel is a ds_read_u8 with an immediate
// offset, and explicit three-operand min/max (80 of them) for the 2 x 16 arcs.
__device__ __forceinline__ int fast_measure_tile(const uint8_t* __restrict__ center, int pitch) {
    const uint8_t* r0 = center - 3;
    const uint8_t* rm1 = r0 - pitch;
    const uint8_t* rm2 = rm1 - pitch;
    const uint8_t* rm3 = rm2 - pitch;
    const uint8_t* rp1 = r0 + pitch;
    const uint8_t* rp2 = rp1 + pitch;
    const uint8_t* rp3 = rp2 + pitch;
    const int v = r0[3];
    int d[16];
    d[0] = v - rp3[3];
    d[1] = v - rp3[4];
    d[2] = v - rp2[5];
    d[3] = v - rp1[6];
    d[4] = v - r0[6];
    d[5] = v - rm1[6];
    d[6] = v - rm2[5];
    d[7] = v - rm3[4];
    d[8] = v - rm3[3];
    d[9] = v - rm3[2];
    d[10] = v - rm2[1];
    d[11] = v - rm1[0];
    d[12] = v - r0[0];
    d[13] = v - rp1[0];
    d[14] = v - rp2[1];
    d[15] = v - rp3[2];
    int mn3[16], mx3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mn3[k] = min3_i32(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
        mx3[k] = max3_i32(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
    }
    int mn9[16], mx9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mn9[k] = min3_i32(mn3[k], mn3[(k + 3) & 15], mn3[(k + 6) & 15]);
        mx9[k] = max3_i32(mx3[k], mx3[(k + 3) & 15], mx3[(k + 6) & 15]);
    }
    // max of the sixteen 9-minima / min of the sixteen 9-maxima: 16 -> 6 -> 2 -> 1
    int a[6], b[6];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        a[k] = max3_i32(mn9[3 * k], mn9[3 * k + 1], mn9[3 * k + 2]);
        b[k] = min3_i32(mx9[3 * k], mx9[3 * k + 1], mx9[3 * k + 2]);
    }
    a[5] = mn9[15];
    b[5] = mx9[15];
    const int A = max(max3_i32(a[0], a[1], a[2]), max3_i32(a[3], a[4], a[5]));
    const int Bn = min(min3_i32(b[0], b[1], b[2]), min3_i32(b[3], b[4], b[5]));
    return max(A, -Bn);
}

}  // namespace orbk
#endif
