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

__device__ __forceinline__ uint32_t max_u16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t min_u16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// The same measure for a pixel of an LDS tile whose row pitch is a run-time value: seven row
// pointers (column -3 of rows -3..3) so that every circle pixel is a ds_read_u8 with an immediate
// offset.  With A = max over arcs of min(v - I_k) = v - min over arcs of max(I_k) and
// B = max over arcs of min(I_k) - v, only the sixteen arc maxima and minima of the raw pixels are
// needed.  An arc of nine starting in one half of the circle is a suffix of that half plus a prefix
// of the other half, so the 2 x 16 arc extrema come from suffix extrema of one half combined with
// a running prefix extremum of the other: 120 two-operand 16-bit min/max, which issue at twice the
// rate of the 32-bit and three-operand forms on this chip.
__device__ __forceinline__ int fast_measure_tile(const uint8_t* __restrict__ center, int pitch) {
    const uint8_t* r0 = center - 3;
    const uint8_t* rm1 = r0 - pitch;
    const uint8_t* rm2 = rm1 - pitch;
    const uint8_t* rm3 = rm2 - pitch;
    const uint8_t* rp1 = r0 + pitch;
    const uint8_t* rp2 = rp1 + pitch;
    const uint8_t* rp3 = rp2 + pitch;
    const int v = r0[3];
    uint32_t p[16];
    p[0] = rp3[3];
    p[1] = rp3[4];
    p[2] = rp2[5];
    p[3] = rp1[6];
    p[4] = r0[6];
    p[5] = rm1[6];
    p[6] = rm2[5];
    p[7] = rm3[4];
    p[8] = rm3[3];
    p[9] = rm3[2];
    p[10] = rm2[1];
    p[11] = rm1[0];
    p[12] = r0[0];
    p[13] = rp1[0];
    p[14] = rp2[1];
    p[15] = rp3[2];
    // arc k = {k .. k+8}: k < 8: suffix of half 0 from k plus prefix of half 1 through k; k >= 8: the
    // mirror.  Per half: its eight suffix extrema, then the other half's prefix extremum runs along.
    uint32_t lo_of_max = 0xffffu, hi_of_min = 0u;
#pragma unroll
    for (int hb = 0; hb < 2; hb++) {
        const uint32_t* q = p + 8 * hb;        // the half the arcs start in
        const uint32_t* o = p + 8 * (hb ^ 1);  // the half they end in
        uint32_t sfx[8];
        sfx[7] = q[7];
#pragma unroll
        for (int i = 6; i >= 0; i--) sfx[i] = max_u16(sfx[i + 1], q[i]);
        uint32_t run = o[0];
        lo_of_max = min_u16(lo_of_max, max_u16(sfx[0], run));
#pragma unroll
        for (int k = 1; k < 8; k++) {
            run = max_u16(run, o[k]);
            lo_of_max = min_u16(lo_of_max, max_u16(sfx[k], run));
        }
        sfx[7] = q[7];
#pragma unroll
        for (int i = 6; i >= 0; i--) sfx[i] = min_u16(sfx[i + 1], q[i]);
        run = o[0];
        hi_of_min = max_u16(hi_of_min, min_u16(sfx[0], run));
#pragma unroll
        for (int k = 1; k < 8; k++) {
            run = min_u16(run, o[k]);
            hi_of_min = max_u16(hi_of_min, min_u16(sfx[k], run));
        }
    }
    return max(v - (int)lo_of_max, (int)hi_of_min - v);
}

}  // namespace orbk
#endif
