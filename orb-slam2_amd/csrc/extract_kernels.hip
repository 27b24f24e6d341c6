// HIP kernels of the ORB extractor for gfx950.  Every kernel works on a batch of frames whose
// pyramids stay resident in HBM; the block index carries (tile or cell or level, frame).
//
// Arithmetic contract (all integer unless stated): results must equal the reference's CPU path
// (src/ORBextractor.cc) with OpenCV 2.4 scalar semantics bit for bit; the two float stages
// (orientation polynomial, pattern rotation) are strict fp32 without contraction, so this file
// is compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "extract_kernels.h"
#include "fast_measure.h"
#include "introselect.h"
#include "wave_select.h"
#include "sincos_port.h"

namespace orbk {

typedef uint32_t u32_unaligned __attribute__((aligned(1)));

__device__ __forceinline__ int reflect101(int p, int n) {
    if (p < 0) p = -p;
    if (p >= n) p = 2 * n - 2 - p;
    return p;
}

// Raw buffer descriptor over `bytes` bytes at p: loads and stores take a 32-bit lane offset plus a scalar offset,
// accesses past the end read zero / are dropped.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
// a * b + c on the low 24 bits of a and b, as ONE instruction (the compiler prefers multiplies and a 3-operand add)
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "s"(a), "v"(b), "v"(c));
    return r;
}
// ---------------------------------------------------------------------------------------------
// Level 0: copyMakeBorder(image, temp, 16.., BORDER_REFLECT_101) (src/ORBextractor.cc:814).
// One thread writes 16 aligned bytes of the padded plane.  The border is 16 px wide, so an interior
// group is a 16-byte run of the source row at a multiple of 16: one 16-byte load when the source
// rows are 16-byte aligned (`wide`), else four (possibly unaligned) dwords; groups that touch the
// border are assembled bytewise through the reflection.
__global__ __launch_bounds__(256) void k_level0(const uint8_t* __restrict__ imgs, long long frame_stride,
                                                long long row_stride, uint8_t* __restrict__ pyr,
                                                const GeomDev* __restrict__ g, int wide, uint32_t gpr_inv) {
    const LevelGeom& L = g->lv[0];
    // the 16-byte groups of the padded plane, row after row, one per thread (a 2-D mapping leaves most lanes of
    // the last block column idle when the row is not a multiple of 512 bytes)
    const int gpr = L.stride >> 4;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const int y = (int)__umulhi(i, gpr_inv);  // i / gpr: gpr_inv = floor(2^32 / gpr) + 1, exact for i < 2^32 / gpr
    const int gx = (int)i - y * gpr;           // 16-byte group of the padded row
    const int f = blockIdx.y;
    if (y >= L.ph) return;
    const uint8_t* src = imgs + (long long)f * frame_stride;
    const int sy = reflect101(y - ORB_EDGE, L.h);
    const uint8_t* srow = src + (long long)sy * row_stride;
    const int x0 = gx * 16 - ORB_EDGE;  // source column of the group's first pixel
    uint4 out;
    if (x0 >= 0 && x0 + 15 < L.w) {
        if (wide) {
            out = *(const uint4*)(srow + x0);
        } else {
            out.x = *(const u32_unaligned*)(srow + x0);
            out.y = *(const u32_unaligned*)(srow + x0 + 4);
            out.z = *(const u32_unaligned*)(srow + x0 + 8);
            out.w = *(const u32_unaligned*)(srow + x0 + 12);
        }
    } else if ((x0 == -ORB_EDGE && L.w > ORB_EDGE) || (x0 == L.w && L.w > ORB_EDGE)) {
        // a group that lies wholly in the left / right border mirrors 16 consecutive source pixels: columns
        // 16 .. 1 for the left border, w - 2 .. w - 17 for the right one -> one 16-byte run, bytes reversed
        const uint8_t* run = srow + (x0 < 0 ? 1 : L.w - 1 - ORB_EDGE);
        const uint32_t a = *(const u32_unaligned*)(run), b = *(const u32_unaligned*)(run + 4),
                       c = *(const u32_unaligned*)(run + 8), d = *(const u32_unaligned*)(run + 12);
        out = make_uint4(__builtin_bswap32(d), __builtin_bswap32(c), __builtin_bswap32(b), __builtin_bswap32(a));
    } else {
        uint32_t wds[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int b = 0; b < 16; b++) {
            const int x = gx * 16 + b;
            uint32_t v = 0;
            if (x < L.w + 2 * ORB_EDGE) v = srow[reflect101(x - ORB_EDGE, L.w)];
            wds[b >> 2] |= v << (8 * (b & 3));
        }
        out = make_uint4(wds[0], wds[1], wds[2], wds[3]);
    }
    *(uint4*)(pyr + (long long)f * g->frame_bytes + L.plane_off + (long long)y * L.stride + gx * 16) = out;
}

// Level l >= 1: resize(level l-1 ROI -> level l ROI, INTER_LINEAR) then copyMakeBorder
// REFLECT_101 (src/ORBextractor.cc:800-807), fused: every padded pixel is computed from the
// interior pixel it reflects to.  Coefficient tables hold OpenCV's 11-bit fixed-point weights.
// One workgroup produces a 256 x 64 pixel tile of the padded destination plane.  The column and
// row maps are monotone up to the border reflection, so the source rectangle of the tile follows
// from the map values at the tile's ends and at the fold points inside it (four wave-uniform
// table reads); that rectangle is staged into LDS with aligned 16-byte loads and the tile is
// interpolated out of LDS.  One thread owns an aligned dword column (4 pixels) of RZ_ROWS rows
// and keeps the horizontal interpolation of the lower source row for the next output row, which
// usually needs it as its upper row.  A wave's rows are the same for all its lanes, so the row
// tables are read through scalar loads.
#define RZ_ROWS 16
#define RZ_TILE_H (4 * RZ_ROWS)

// extremes of reflect101(t, n) over t in [a, b] (a <= b): (min, max)
__device__ __forceinline__ void reflect_range(int a, int b, int n, int& lo, int& hi) {
    const int ra = reflect101(a, n), rb = reflect101(b, n);
    lo = (a <= 0 && b >= 0) ? 0 : min(ra, rb);
    hi = (a <= n - 1 && b >= n - 1) ? n - 1 : max(ra, rb);
}

// Rows [ybase, yend) of one dword column: vertical pass of cv::resize over horizontally
// interpolated source rows read from `rows` (row r at rows + (r - row0) * rpitch).
// "((b0 * (T0 >> 4)) >> 16) + ((b1 * (T1 >> 4)) >> 16) + 2) >> 2": the two products are taken as
// the high halves of (b << 16) * (T >> 4).
__device__ __forceinline__ void resize_rows(const uint8_t* __restrict__ rows, int rpitch, int row0, const int sx[4],
                                            const int sx1[4], const int a0[4], const int a1[4],
                                            const uint32_t* __restrict__ ytab_s, const uint32_t* __restrict__ ytab_b,
                                            int ybase, int yend, int h, __amdgpu_buffer_rsrc_t dst, uint32_t dst_off,
                                            int dstride, uint32_t keep) {
    int cached_row = -1;
    uint32_t ct[4] = {0, 0, 0, 0};
#pragma unroll 2
    for (int y = ybase; y < yend; y++) {
        const int dy = reflect101(y - ORB_EDGE, h);
        const uint32_t ys = ytab_s[dy], yb = ytab_b[dy];
        const int sy0 = ys & 0xffff, sy1 = ys >> 16;
        const uint32_t b0s = yb << 16, b1s = yb & 0xffff0000u;
        const int o1 = (sy1 - row0) * rpitch;
        uint32_t t0[4], t1[4];
        if (sy0 == cached_row) {
#pragma unroll
            for (int b = 0; b < 4; b++) t0[b] = ct[b];
        } else {
            const int o0 = (sy0 - row0) * rpitch;
#pragma unroll
            for (int b = 0; b < 4; b++) t0[b] = (uint32_t)(rows[o0 + sx[b]] * a0[b] + rows[o0 + sx1[b]] * a1[b]) >> 4;
        }
#pragma unroll
        for (int b = 0; b < 4; b++) t1[b] = (uint32_t)(rows[o1 + sx[b]] * a0[b] + rows[o1 + sx1[b]] * a1[b]) >> 4;
        uint32_t out = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t v = (__umulhi(b0s, t0[b]) + __umulhi(b1s, t1[b]) + 2u) >> 2;
            out |= v << (8 * b);
            ct[b] = t1[b];
        }
        cached_row = sy1;
        __builtin_amdgcn_raw_buffer_store_b32(out & keep, dst, dst_off, y * dstride, 0);  // row offset rides in a scalar register
    }
}

// Developer instrumentation (-DORBHIP_RZ_PHASE_TIMING, never in a shipped build): shader-clock cycles the waves of
// k_resize spend in each phase, summed; read and reset through orbhip_debug_rz_phase.
#ifdef ORBHIP_RZ_PHASE_TIMING
__device__ unsigned long long rz_phase[8];
#define RZ_T0 unsigned long long rzt_ = __builtin_readcyclecounter();
#define RZ_T(k)                                                                           \
    {                                                                                     \
        const unsigned long long now_ = __builtin_readcyclecounter();                     \
        if (threadIdx.x == 0) atomicAdd(&rz_phase[k], now_ - rzt_);                       \
        rzt_ = now_;                                                                      \
    }
#else
#define RZ_T0
#define RZ_T(k)
#endif

__global__ __launch_bounds__(256) void k_resize(uint8_t* __restrict__ pyr, const GeomDev* __restrict__ g, int level,
                                                const int* __restrict__ xofs, const uint32_t* __restrict__ xab,
                                                const uint32_t* __restrict__ ysrc, const uint32_t* __restrict__ yab,
                                                int lds_pitch, int lds_rows) {
    extern __shared__ uint4 rz_lds[];
    RZ_T0
    const LevelGeom& L = g->lv[level];
    const LevelGeom& P = g->lv[level - 1];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    int bx, by, f;
    xcd_frame_swizzle3(bx, by, f);
    const int dwx = bx * 64 + threadIdx.x;
    const int wy = __builtin_amdgcn_readfirstlane((int)threadIdx.y);  // one wave per threadIdx.y: wave-uniform rows
    const int ybase = (by * 4 + wy) * RZ_ROWS;
    uint8_t* frame = pyr + (long long)f * g->frame_bytes;
    const uint8_t* src = frame + P.plane_off + (long long)ORB_EDGE * P.stride + ORB_EDGE;
    const bool col_ok = dwx * 4 < L.stride;
    const int pw = L.w + 2 * ORB_EDGE;
    // source rectangle of the whole tile (workgroup-uniform)
    int dlo, dhi;
    reflect_range(bx * 256 - ORB_EDGE, min(bx * 256 + 255, pw - 1) - ORB_EDGE, L.w, dlo, dhi);
    const int X0 = xofs[L.rz_x_off + dlo] & ~15;
    const int Xmax = min(xofs[L.rz_x_off + dhi] + 1, P.w - 1);
    reflect_range(by * RZ_TILE_H - ORB_EDGE, min(by * RZ_TILE_H + RZ_TILE_H, L.ph) - 1 - ORB_EDGE, L.h, dlo, dhi);
    const int sy_min = ysrc[L.rz_y_off + dlo] & 0xffff;
    const int nrows = (int)(ysrc[L.rz_y_off + dhi] >> 16) - sy_min + 1;
    const int pc = (Xmax - X0 + 16) >> 4;  // 16-byte chunks per staged row
    // host sized the LDS for the level's scale; a tile that would not fit reads global memory
    const bool staged = pc * 16 <= lds_pitch && nrows <= lds_rows && pc <= 32;
    RZ_T(0)
    if (staged) {
        // 32 lanes per source row, 8 rows per sweep of the workgroup, four rows in flight per lane
        const int cj = tid & 31, r0 = tid >> 5;
        if (cj < pc) {
            const int lp4 = lds_pitch >> 4;
            const uint8_t* s0 = src + (long long)sy_min * P.stride + X0 + 16 * cj;
            for (int r = r0; r < nrows; r += 32) {
                uint4 v[4];
                int rr[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    rr[k] = min(r + 8 * k, nrows - 1);  // clamped rows store the same bytes again
                    v[k] = *(const uint4*)(s0 + (long long)rr[k] * P.stride);
                }
#pragma unroll
                for (int k = 0; k < 4; k++) rz_lds[rr[k] * lp4 + cj] = v[k];
            }
        }
    }
    RZ_T(1)
    int sx[4], sx1[4], a0[4], a1[4];
    uint32_t keep = 0;  // bytes of the dword that are inside the padded plane
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int x = min(dwx * 4 + b, pw - 1);
        if (dwx * 4 + b < pw) keep |= 0xffu << (8 * b);
        const int dx = reflect101(x - ORB_EDGE, L.w);
        sx[b] = xofs[L.rz_x_off + dx];
        sx1[b] = min(sx[b] + 1, P.w - 1);
        const uint32_t xa = xab[L.rz_x_off + dx];
        a0[b] = xa & 0xffff;
        a1[b] = xa >> 16;
    }
    const int yend = min(ybase + RZ_ROWS, L.ph);
    RZ_T(2)
    __syncthreads();
    RZ_T(3)
    if (!col_ok || ybase >= yend) return;
    const __amdgpu_buffer_rsrc_t dst = plane_rsrc(frame + L.plane_off, L.stride * L.ph);
    const uint32_t dst_off = (uint32_t)dwx * 4u;
    const uint32_t* ytab_s = ysrc + L.rz_y_off;
    const uint32_t* ytab_b = yab + L.rz_y_off;
    // offsets are formed as non-negative ints and added to the tile / plane origin: no pointer ever
    // leaves its buffer (LDS pointers are 32-bit and must not wrap)
    if (staged) {
#pragma unroll
        for (int b = 0; b < 4; b++) {
            sx[b] -= X0;
            sx1[b] -= X0;
        }
        resize_rows((const uint8_t*)rz_lds, lds_pitch, sy_min, sx, sx1, a0, a1, ytab_s, ytab_b, ybase, yend, L.h, dst,
                    dst_off, L.stride, keep);
        RZ_T(4)
    } else {
        resize_rows(src, P.stride, 0, sx, sx1, a0, a1, ytab_s, ytab_b, ybase, yend, L.h, dst, dst_off, L.stride, keep);
    }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_score(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ score,
                                               const GeomDev* __restrict__ g, int level) {
    const LevelGeom& L = g->lv[level];
    const int x = ORB_EDGE + blockIdx.x * 64 + threadIdx.x;
    const int y = ORB_EDGE + blockIdx.y * 4 + threadIdx.y;
    const int f = blockIdx.z;
    if (x >= L.w - ORB_EDGE || y >= L.h - ORB_EDGE) return;
    const long long off = (long long)f * g->frame_bytes + L.plane_off + (long long)(y + ORB_EDGE) * L.stride + (x + ORB_EDGE);
    int m = fast_measure(pyr + off, L.stride);
    score[off] = (uint8_t)min(max(m, 0), 255);
}

// ---------------------------------------------------------------------------------------------
// Per-cell non-max suppression, adaptive threshold and raster-ordered compaction
// (src/ORBextractor.cc:599-614 + cv::FAST's NMS).  One workgroup per (cell, frame).
// A pixel is kept at threshold t iff M > t and its score M-1 is strictly greater than the
// score of each of its 8 neighbours, where a neighbour that is not a corner at t, or lies
// outside the pixels this cell's FAST call examines, counts as 0.
struct NmsCtx {
    const uint8_t* S;  // measure plane at the level ROI origin
    int stride;
    int x0, y0, x1, y1;
};

__device__ __forceinline__ void nms_flags(const NmsCtx& c, int x, int y, int tHi, int tLo, bool& fHi, bool& fLo) {
    const uint8_t* p = c.S + (long long)y * c.stride + x;
    const int m = p[0];
    fHi = m > tHi;
    fLo = m > tLo;
    if (!fHi && !fLo) return;
    const int s = m - 1;
    int nbHi = 0, nbLo = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            if (dx == 0 && dy == 0) continue;
            const int xx = x + dx, yy = y + dy;
            if (xx < c.x0 || xx >= c.x1 || yy < c.y0 || yy >= c.y1) continue;
            const int q = p[dy * c.stride + dx];
            nbHi = max(nbHi, q > tHi ? q - 1 : 0);
            nbLo = max(nbLo, q > tLo ? q - 1 : 0);
        }
    fHi = fHi && s > nbHi;
    fLo = fLo && s > nbLo;
}

__global__ __launch_bounds__(256) void k_nms(const uint8_t* __restrict__ score, const GeomDev* __restrict__ g,
                                             const CellDesc* __restrict__ cells, uint32_t* __restrict__ cand,
                                             int* __restrict__ n_total, int* __restrict__ err) {
    extern __shared__ int sh[];  // [0,ch): counts/offsets at fastTh, [ch,2ch): at 7, then 2*256 partials + 2
    const int c = blockIdx.x, f = blockIdx.y;
    const CellDesc cd = cells[c];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    int* out_total = n_total + (long long)f * g->cells_per_frame + c;
    if (cd.cw <= 0 || cd.ch <= 0) {
        if (tid == 0) *out_total = 0;
        return;
    }
    const LevelGeom& L = g->lv[cd.level];
    NmsCtx ctx;
    ctx.S = score + (long long)f * g->frame_bytes + L.plane_off + (long long)ORB_EDGE * L.stride + ORB_EDGE;
    ctx.stride = L.stride;
    ctx.x0 = cd.x0;
    ctx.y0 = cd.y0;
    ctx.x1 = cd.x0 + cd.cw;
    ctx.y1 = cd.y0 + cd.ch;
    const int ch = cd.ch, cw = cd.cw;
    const int tHi = min(max(g->fast_th, 0), 255), tLo = 7;
    int* cntHi = sh;
    int* cntLo = sh + ch;
    int* part = sh + 2 * ch;  // 512 partial sums, then 2 totals

    for (int r = wave; r < ch; r += 4) {
        int nHi = 0, nLo = 0;
        for (int xb = 0; xb < cw; xb += 64) {
            bool fHi = false, fLo = false;
            if (xb + lane < cw) nms_flags(ctx, cd.x0 + xb + lane, cd.y0 + r, tHi, tLo, fHi, fLo);
            nHi += __popcll(__ballot(fHi));
            nLo += __popcll(__ballot(fLo));
        }
        if (lane == 0) {
            cntHi[r] = nHi;
            cntLo[r] = nLo;
        }
    }
    __syncthreads();
    // exclusive scan over rows: thread t owns rows [t*k, (t+1)*k)
    const int k = (ch + 255) / 256;
    const int r0 = min(tid * k, ch), r1 = min(r0 + k, ch);
    int sHi = 0, sLo = 0;
    for (int r = r0; r < r1; r++) {
        sHi += cntHi[r];
        sLo += cntLo[r];
    }
    part[tid] = sHi;
    part[256 + tid] = sLo;
    __syncthreads();
    if (tid == 0) {
        int aHi = 0, aLo = 0;
        for (int t = 0; t < 256; t++) {
            int vHi = part[t], vLo = part[256 + t];
            part[t] = aHi;
            part[256 + t] = aLo;
            aHi += vHi;
            aLo += vLo;
        }
        part[512] = aHi;
        part[513] = aLo;
    }
    __syncthreads();
    {
        int aHi = part[tid], aLo = part[256 + tid];
        for (int r = r0; r < r1; r++) {
            int vHi = cntHi[r], vLo = cntLo[r];
            cntHi[r] = aHi;
            cntLo[r] = aLo;
            aHi += vHi;
            aLo += vLo;
        }
    }
    __syncthreads();
    const bool useLo = part[512] <= 3;  // "if(cellKeyPoints[i][j].size()<=3)" -> FAST(..., 7, true)
    int total = useLo ? part[513] : part[512];
    const int* offs = useLo ? cntLo : cntHi;
    uint32_t* out = cand + (long long)f * g->cand_per_frame + cd.cand_off;
    if (total > cd.cand_cap) {
        if (tid == 0) atomicOr(err, 1);
        total = cd.cand_cap;
    }
    for (int r = wave; r < ch; r += 4) {
        int off = offs[r];
        for (int xb = 0; xb < cw; xb += 64) {
            bool fHi = false, fLo = false;
            const int x = cd.x0 + xb + lane, y = cd.y0 + r;
            if (xb + lane < cw) nms_flags(ctx, x, y, tHi, tLo, fHi, fLo);
            const bool fl = useLo ? fLo : fHi;
            const unsigned long long mask = __ballot(fl);
            if (fl) {
                const int pos = off + __popcll(mask & ((1ull << lane) - 1ull));
                if (pos < cd.cand_cap) {
                    const uint32_t s = (uint32_t)ctx.S[(long long)y * ctx.stride + x] - 1u;
                    out[pos] = (s << 24) | ((uint32_t)(y - cd.ini_y) << 12) | (uint32_t)(x - cd.ini_x);
                }
            }
            off += __popcll(mask);
        }
    }
    if (tid == 0) *out_total = total;
}

// ---------------------------------------------------------------------------------------------
// HarrisResponses(cellImage, keypoints, 7, HARRIS_K) (src/ORBextractor.cc:79-120,616-620): replaces
// the FAST score of every candidate by the Harris measure of the 7 x 7 block around it.  One lane
// per candidate reads the 9 x 9 window once (rows of three unaligned dwords) and evaluates the
// reference's expression operation by operation in fp32.  The response goes into bits 63..32 of a
// 64-bit key as an unsigned integer with the same order as the float; the 32-bit key stays below.
__device__ __forceinline__ uint32_t float_to_ordered(float f) {
    if (f == 0.f) f = 0.f;  // -0 and +0 compare equal as floats
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ __launch_bounds__(256) void k_harris(const uint8_t* __restrict__ pyr, const GeomDev* __restrict__ g,
                                                const CellDesc* __restrict__ cells, const uint32_t* __restrict__ cand,
                                                const int* __restrict__ n_total, uint64_t* __restrict__ cand64) {
    const int c = blockIdx.x, f = blockIdx.y;
    const CellDesc cd = cells[c];
    const int n = min(n_total[(long long)f * g->cells_per_frame + c], cd.cand_cap);
    const LevelGeom& L = g->lv[cd.level];
    const int stride = L.stride;
    const uint8_t* plane = pyr + (long long)f * g->frame_bytes + L.plane_off;
    const uint32_t* src = cand + (long long)f * g->cand_per_frame + cd.cand_off;
    uint64_t* dst = cand64 + (long long)f * g->cand_per_frame + cd.cand_off;
    float scale = (1 << 2) * 7 * 255.0f;
    scale = 1.0f / scale;
    const float scale_sq_sq = __fmul_rn(__fmul_rn(__fmul_rn(scale, scale), scale), scale);
    const float harris_k = 0.04f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const uint32_t key = src[i];
        const int x = key & 0xfff, y = (key >> 12) & 0xfff;  // cell-image coordinates
        // window rows y-4 .. y+4, columns x-4 .. x+4 in padded-plane coordinates
        const uint8_t* w0 = plane + (long long)(ORB_EDGE + cd.ini_y + y - 4) * stride + (ORB_EDGE + cd.ini_x + x - 4);
        int px[9][9];
#pragma unroll
        for (int r = 0; r < 9; r++) {
            const uint8_t* row = w0 + (long long)r * stride;
            const uint32_t d0 = *(const u32_unaligned*)row, d1 = *(const u32_unaligned*)(row + 4);
            const uint32_t d2 = row[8];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                px[r][k] = (d0 >> (8 * k)) & 0xff;
                px[r][4 + k] = (d1 >> (8 * k)) & 0xff;
            }
            px[r][8] = d2;
        }
        int a = 0, b = 0, cc = 0;
#pragma unroll
        for (int r = 1; r < 8; r++)
#pragma unroll
            for (int k = 1; k < 8; k++) {
                const int Ix = (px[r][k + 1] - px[r][k - 1]) * 2 + (px[r - 1][k + 1] - px[r - 1][k - 1]) +
                               (px[r + 1][k + 1] - px[r + 1][k - 1]);
                const int Iy = (px[r + 1][k] - px[r - 1][k]) * 2 + (px[r + 1][k - 1] - px[r - 1][k - 1]) +
                               (px[r + 1][k + 1] - px[r - 1][k + 1]);
                a += Ix * Ix;
                b += Iy * Iy;
                cc += Ix * Iy;
            }
        const float fa = (float)a, fb = (float)b, fc = (float)cc;
        const float sum = __fadd_rn(fa, fb);
        const float resp = __fmul_rn(
            __fsub_rn(__fsub_rn(__fmul_rn(fa, fb), __fmul_rn(fc, fc)), __fmul_rn(__fmul_rn(harris_k, sum), sum)), scale_sq_sq);
        dst[i] = ((uint64_t)float_to_ordered(resp) << 32) | key;
    }
}

// ---------------------------------------------------------------------------------------------
// Quota redistribution (src/ORBextractor.cc:622-670), per-cell retainBest (:683-685), cell-major
// concatenation in level coordinates (:687-694) and per-level retainBest (:697-701), as three
// kernels so that the per-cell selections (the bulk of the work, one wave each) spread over the
// whole chip instead of queueing behind the four waves of their level's workgroup:
//   k_quota         one workgroup per (level, frame).  Every pass of the quota loop over the cells
//                   is order-independent (sums and per-cell decisions), so it runs data-parallel
//                   with block reductions; the while loop itself is the reference's.
//   k_cell_select   one wave per (cell, frame): list staged in LDS, exact std::nth_element element
//                   movement with wave-parallel partitions (wave_select.h); the permuted list goes
//                   back to the cell's arena.
//   k_level_select  one workgroup per (level, frame): exclusive scan of nToRetain -> offsets,
//                   gather of the retained prefixes in level coordinates, per-level selection by
//                   one wave, level list and count out.
// Cell lists up to SEL_WCAP keys are staged in LDS; up to twice that the keys stay in global memory
// and only the stop lists use LDS; larger lists (and level lists above SEL_LCAP) are selected in
// place in global memory by one lane (introselect.h) — same results, only slower.  SEL_WCAP sets the
// LDS per wave (4 KB at 512 keys) and with it the waves in flight of this latency-bound kernel.
#define SEL_WCAP 512
#define SEL_LCAP 4096

__device__ __forceinline__ int block_sum(int v, int* red, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void k_quota(const GeomDev* __restrict__ g, const CellDesc* __restrict__ cells,
                                               const int* __restrict__ n_total, int* __restrict__ n_retain) {
    __shared__ int ltot[ORB_MAX_CELLS_PER_LEVEL];
    __shared__ int lret[ORB_MAX_CELLS_PER_LEVEL];
    __shared__ uint8_t done[ORB_MAX_CELLS_PER_LEVEL];
    __shared__ int red[4];
    const int level = blockIdx.x, f = blockIdx.y;
    const LevelGeom& L = g->lv[level];
    const int tid = threadIdx.x;
    const int nCells = L.nCells;
    const CellDesc* lc = cells + L.cell_base;
    const int* tot = n_total + (long long)f * g->cells_per_frame + L.cell_base;
    int* ret = n_retain + (long long)f * g->cells_per_frame + L.cell_base;
    const int nfc = L.nfeaturesCell;
    int d = 0, closed = 0;
    for (int c = tid; c < nCells; c += 256) {
        // a visited cell with nothing to examine closes here with zero keys; only a cell the
        // reference never visits stays open for the loop below
        const bool skipped = lc[c].skipped != 0;
        const int nKeys = tot[c];
        ltot[c] = nKeys;
        if (skipped) {
            lret[c] = 0;
            done[c] = 0;
        } else if (nKeys > nfc) {
            lret[c] = nfc;
            done[c] = 0;
        } else {
            lret[c] = nKeys;
            d += nfc - nKeys;
            done[c] = 1;
            closed++;
        }
    }
    int nToDistribute = block_sum(d, red, tid);
    int nNoMore = block_sum(closed, red, tid);
    while (nToDistribute > 0 && nNoMore < nCells) {
        const int nNew = nfc + (int)ceilf((float)nToDistribute / (float)(nCells - nNoMore));
        d = 0;
        closed = 0;
        for (int c = tid; c < nCells; c += 256)
            if (!done[c]) {
                const int nt = ltot[c];
                if (nt > nNew) {
                    lret[c] = nNew;
                } else {
                    lret[c] = nt;
                    d += nNew - nt;
                    done[c] = 1;
                    closed++;
                }
            }
        nToDistribute = block_sum(d, red, tid);
        nNoMore += block_sum(closed, red, tid);
    }
    __syncthreads();
    for (int c = tid; c < nCells; c += 256) ret[c] = lret[c];
}

template <class KeyT>
__global__ __launch_bounds__(256) void k_cell_select(const GeomDev* __restrict__ g, const CellDesc* __restrict__ cells,
                                                     KeyT* __restrict__ cand, const int* __restrict__ n_total,
                                                     const int* __restrict__ n_retain) {
    __shared__ KeyT buf[4 * SEL_WCAP];
    __shared__ uint16_t lists[4 * 2 * SEL_WCAP];
    int bx = blockIdx.x, f = blockIdx.y;
    xcd_frame_swizzle(bx, f, (int)gridDim.x, (int)gridDim.y);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int c = bx * 4 + wave;
    if (c >= g->cells_per_frame) return;
    const long long fc = (long long)f * g->cells_per_frame + c;
    const int n = n_total[fc], kk = n_retain[fc];
    if (!(n > kk && kk > 0)) return;
    KeyT* list = cand + (long long)f * g->cand_per_frame + cells[c].cand_off;
    if (n <= SEL_WCAP) {
        KeyT* wbuf = buf + wave * SEL_WCAP;
        uint16_t* wl = lists + wave * 2 * SEL_WCAP;
        for (int i = lane; i < n; i += 64) wbuf[i] = list[i];
        orbsel::wave_lds_fence();
        orbsel::wave_nth_element_desc(wbuf, n, kk, wl, wl + SEL_WCAP, lane);
        // the whole permuted list goes back, so the cell list stays a permutation of the candidates
        for (int i = lane; i < n; i += 64) list[i] = wbuf[i];
    } else if (n <= 2 * SEL_WCAP) {
        // keys stay in the cell's arena (cached global memory); the wave's whole LDS share holds the two
        // stop lists: same wave-cooperative rounds, slower element access
        uint16_t* lp = lists + wave * 2 * SEL_WCAP;           // 2 * SEL_WCAP entries
        uint16_t* rp = (uint16_t*)(buf + wave * SEL_WCAP);    // SEL_WCAP keys = at least 2 * SEL_WCAP entries
        orbsel::wave_nth_element_desc(list, n, kk, lp, rp, lane);
    } else if (lane == 0) {
        orbsel::nth_element_desc(list, n, kk);
    }
}

template <class KeyT>
__global__ __launch_bounds__(256) void k_level_select(const GeomDev* __restrict__ g, const CellDesc* __restrict__ cells,
                                                      const KeyT* __restrict__ cand, const int* __restrict__ n_retain,
                                                      KeyT* __restrict__ lvl_list, int* __restrict__ lvl_count,
                                                      int* __restrict__ err, int lcap) {
    // staging area for level lists of up to lcap keys (keys, then the two stop lists), sized by the host from the
    // largest level list this geometry can produce: the workgroups a CU holds follow from it
    extern __shared__ uint4 lsel_lds[];
    KeyT* buf = (KeyT*)lsel_lds;
    uint16_t* lists = (uint16_t*)(buf + lcap);
    __shared__ int aux[ORB_MAX_CELLS_PER_LEVEL + 1];
    __shared__ int red[4];
    const int level = blockIdx.x, f = blockIdx.y;
    const LevelGeom& L = g->lv[level];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nCells = L.nCells;
    const CellDesc* lc = cells + L.cell_base;
    const int* ret = n_retain + (long long)f * g->cells_per_frame + L.cell_base;
    const KeyT* fcand = cand + (long long)f * g->cand_per_frame;

    // ---- exclusive scan of nToRetain over the cells (consecutive cells per thread) ----------------
    {
        const int per = (nCells + 255) / 256;
        const int c0 = min(tid * per, nCells), c1 = min(c0 + per, nCells);
        int s = 0;
        for (int c = c0; c < c1; c++) s += ret[c];
        int incl = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) red[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int w2 = 0; w2 < wave; w2++) woff += red[w2];
        const int grand = red[0] + red[1] + red[2] + red[3];
        int run = woff + incl - s;
        for (int c = c0; c < c1; c++) {
            aux[c] = run;
            run += ret[c];
        }
        if (tid == 0) aux[nCells] = grand;
        __syncthreads();
    }
    int total = aux[nCells];
    KeyT* ll = lvl_list + (long long)f * g->lvl_per_frame + L.lvl_off;
    if (total > L.lvl_cap) {
        if (tid == 0) atomicOr(err, 2);
        total = L.lvl_cap;
    }
    const bool lstaged = total <= lcap;
    KeyT* dst = lstaged ? buf : ll;
    for (int c = wave; c < nCells; c += 4) {
        const int o = aux[c], r = aux[c + 1] - o;
        const KeyT* list = fcand + lc[c].cand_off;
        const uint32_t add = ((uint32_t)lc[c].ini_y << 12) + (uint32_t)lc[c].ini_x;
        for (int i = lane; i < r; i += 64)
            if (o + i < total) dst[o + i] = list[i] + add;  // x, y of the cell image -> level coordinates
    }
    __threadfence_block();
    __syncthreads();

    // ---- per-level selection ------------------------------------------------------------------------
    const int quota = L.quota;
    if (total > quota) {
        if (lstaged) {
            if (wave == 0) orbsel::wave_nth_element_desc(buf, total, quota, lists, lists + lcap, lane);
        } else if (tid == 0) {
            orbsel::nth_element_desc(ll, total, quota);
        }
    }
    __syncthreads();
    const int count = min(total, quota);
    if (lstaged)
        for (int i = tid; i < count; i += 256) ll[i] = buf[i];
    if (tid == 0) lvl_count[f * g->nlevels + level] = count;
}

// ---------------------------------------------------------------------------------------------
// Output slot s of a frame belongs to level l iff slot_base[l] <= s < slot_base[l] + quota[l];
// k = s - slot_base[l] is the keypoint's rank inside the level list.
// key_words = 1: 32-bit keys; 2: 64-bit keys (response bits in the upper word, returned in `hi`).
__device__ __forceinline__ bool slot_to_key(const GeomDev* g, const uint32_t* lvl_list, int key_words, const int* lvl_count,
                                            int f, int slot, int& level, int& k, uint32_t& key, uint32_t& hi,
                                            int& out_index) {
    if (slot >= g->nfeatures) return false;
    int l = 0;
    while (l + 1 < g->nlevels && g->lv[l + 1].slot_base <= slot) l++;
    k = slot - g->lv[l].slot_base;
    const int* cnt = lvl_count + f * g->nlevels;
    if (k >= cnt[l]) return false;
    int o = 0;
    for (int i = 0; i < l; i++) o += cnt[i];
    out_index = o + k;
    level = l;
    const long long e = ((long long)f * g->lvl_per_frame + g->lv[l].lvl_off + k) * key_words;
    key = lvl_list[e];
    hi = key_words == 2 ? lvl_list[e + 1] : 0u;
    return true;
}

// cv::fastAtan2 of OpenCV 2.4 (degrees), strict fp32.
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale,
                p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, ax + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = __fdiv_rn(ax, ay + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// ---------------------------------------------------------------------------------------------
// GaussianBlur(level, level, Size(7,7), 2, 2, BORDER_REFLECT_101) on the level ROI
// (src/ORBextractor.cc:760): separable integer taps [18,34,49,55,49,34,18], row pass in int32,
// column pass (sum + 2^15) >> 16.  The filter border is the real 3-px ring of the padded plane.
// Output goes to a second plane whose 16-px margin is a copy of the unblurred margin: the in-place
// reference leaves the margin unblurred and the descriptor pattern reaches up to 2 px into it.
//
// One thread owns a column of 4 pixels (one aligned dword) and walks BL_ROWS rows downwards.  Per
// input row it loads three dwords (coalesced 256 B per wave), forms the four 7-tap row sums with
// ten v_dot4_u32_u8 against tap words laid out per pixel, and keeps the last seven rows of sums in registers (the
// row loop is fully unrolled so the ring slots are static); per output row the column pass is
// 7 integer ops per pixel.  No LDS: neighbouring lanes' overlapping loads hit the vector L1.
#define BL_ROWS 32
#define BL_PREFETCH 8

// TIE_EVEN selects how an exact tie of the column pass (sum mod 2^16 == 2^15) rounds in the first w & ~3 ROI
// columns: false = half up, (sum + 2^15) >> 16, the scalar column filter of OpenCV 2.4; true = to the even
// neighbour, which is what the fp32 + cvtps2dq column filter of an SSE2 build (SymmColumnVec_32s8u) produces
// there — every product and partial sum of that filter is exact in fp32 below saturation, so ties are the only
// difference.
// All levels of the batch are blurred by ONE launch: the tiles of the levels are numbered one level after the
// other (BlurPlan), so the small upper levels cost neither a launch of their own nor a tail in which most of the
// chip waits for their few workgroups.
struct BlurPlan {
    int first[ORB_MAX_LEVELS + 1];  // first tile of each level; first[nlevels] = tiles per frame
    int nx[ORB_MAX_LEVELS];         // tile columns of each level
    int nlevels;
};
template <bool TIE_EVEN>
__global__ __launch_bounds__(256) void k_blur(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur,
                                              const GeomDev* __restrict__ g, BlurPlan plan, int t0, int t1, int t2, int t3) {
    int tile = blockIdx.x, f = blockIdx.y;
    xcd_frame_swizzle(tile, f, (int)gridDim.x, (int)gridDim.y);
    int level = 0;
    while (level + 1 < plan.nlevels && tile >= plan.first[level + 1]) level++;
    tile -= plan.first[level];
    const int by = tile / plan.nx[level], bx = tile - by * plan.nx[level];
    const LevelGeom& L = g->lv[level];
    const int ndw = L.stride >> 2;
    const int dwx = bx * 64 + threadIdx.x;
    // one wave per threadIdx.y: the rows a wave walks are the same for all its lanes (scalar row arithmetic)
    const int y0 = (by * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.y)) * BL_ROWS;
    const int ph = L.ph;
    if (dwx >= ndw || y0 >= ph) return;
    const long long base = (long long)f * g->frame_bytes + L.plane_off;
    const uint8_t* src = pyr + base;
    uint8_t* dst = blur + base;
    const int stride = L.stride;
    const int xl = max(dwx - 1, 0), xr = min(dwx + 1, ndw - 1);
    // tap words: byte i of T?j multiplies byte i of the dword (taps 0..6 = t0 t1 t2 t3 t2 t1 t0)
#define TAPW(b0, b1, b2, b3) ((uint32_t)(b0) | ((uint32_t)(b1) << 8) | ((uint32_t)(b2) << 16) | ((uint32_t)(b3) << 24))
    const uint32_t TA0 = TAPW(0, t0, t1, t2), TB0 = TAPW(t3, t2, t1, t0);
    const uint32_t TA1 = TAPW(0, 0, t0, t1), TB1 = TAPW(t2, t3, t2, t1), TC1 = TAPW(t0, 0, 0, 0);
    const uint32_t TA2 = TAPW(0, 0, 0, t0), TB2 = TAPW(t1, t2, t3, t2), TC2 = TAPW(t1, t0, 0, 0);
    const uint32_t TB3 = TAPW(t0, t1, t2, t3), TC3 = TAPW(t2, t1, t0, 0);
#undef TAPW
    // pixels of this dword inside the ROI columns [16, 16 + w)
    const int px0 = dwx * 4;
    uint32_t colmask = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (px0 + j >= ORB_EDGE && px0 + j < L.w + ORB_EDGE) colmask |= 0xffu << (8 * j);
    uint32_t h[7][4];
    uint32_t ctr[7];
    const uint32_t offl = (uint32_t)xl * 4u, offm = (uint32_t)dwx * 4u, offr = (uint32_t)xr * 4u;
    const __amdgpu_buffer_rsrc_t rsrc = plane_rsrc(src, stride * ph), rdst = plane_rsrc(dst, stride * ph);
#pragma unroll
    for (int c0 = 0; c0 < BL_ROWS + 6; c0 += BL_PREFETCH) {
        // issue the loads of BL_PREFETCH rows back to back, then do their arithmetic
        uint32_t ra[BL_PREFETCH], rb[BL_PREFETCH], rc[BL_PREFETCH];
#pragma unroll
        for (int k = 0; k < BL_PREFETCH; k++)
            if (c0 + k < BL_ROWS + 6) {
                // buffer loads: wave-uniform row offset in a scalar register, 32-bit lane offset, no per-lane
                // 64-bit address arithmetic
                const int yy = min(max(y0 - 3 + c0 + k, 0), ph - 1);
                ra[k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, offl, yy * stride, 0);
                rb[k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, offm, yy * stride, 0);
                rc[k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, offr, yy * stride, 0);
            }
#pragma unroll
        for (int k = 0; k < BL_PREFETCH; k++)
            if (c0 + k < BL_ROWS + 6) {
                const int i = c0 + k;
                const uint32_t a = ra[k], b = rb[k], c = rc[k];
                const int slot = i % 7;
                ctr[slot] = b;
                // bytes of (a, b, c) are columns px0-4 .. px0+7; pixel j needs byte offsets j+1 .. j+7.  The dwords
                // stay as loaded and the taps are placed where each pixel's window falls in them (zeros
                // elsewhere): ten dot products per four pixels and no byte alignment
                h[slot][0] = __builtin_amdgcn_udot4(a, TA0, __builtin_amdgcn_udot4(b, TB0, 0u, false), false);
                h[slot][1] = __builtin_amdgcn_udot4(a, TA1, __builtin_amdgcn_udot4(b, TB1, __builtin_amdgcn_udot4(c, TC1, 0u, false), false), false);
                h[slot][2] = __builtin_amdgcn_udot4(a, TA2, __builtin_amdgcn_udot4(b, TB2, __builtin_amdgcn_udot4(c, TC2, 0u, false), false), false);
                h[slot][3] = __builtin_amdgcn_udot4(b, TB3, __builtin_amdgcn_udot4(c, TC3, 0u, false), false);
                if (i >= 6) {
                    const int y = y0 + i - 6;
                    if (y < ph) {
                        uint32_t o = ctr[(i - 3) % 7];
                        if (y >= ORB_EDGE && y < L.h + ORB_EDGE && colmask) {
                            // (sum + 2^15) >> 16 saturated to 255: the rounding constant rides on the centre tap's
                            // multiply-add, the saturation is a clamp to 2^24 - 1, and byte 2 of each clamped sum
                            // is the pixel (two byte permutes assemble the four of them)
                            uint32_t acc[4];
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                // row sums are below 2^17 and taps below 2^8: 24-bit multiply-adds
                                uint32_t a = mad_u24((uint32_t)t3, h[(i - 3) % 7][j], 32768u);
                                a = mad_u24((uint32_t)t0, h[(i - 6) % 7][j] + h[i % 7][j], a);
                                a = mad_u24((uint32_t)t1, h[(i - 5) % 7][j] + h[(i - 1) % 7][j], a);
                                a = mad_u24((uint32_t)t2, h[(i - 4) % 7][j] + h[(i - 2) % 7][j], a);
                                if (TIE_EVEN && px0 + j - ORB_EDGE < (L.w & ~3) && (a & 0xffffu) == 0u) a &= ~0x10000u;
                                acc[j] = min(a, 0xffffffu);
                            }
                            const uint32_t bl = __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0602u) |
                                                __builtin_amdgcn_perm(acc[3], acc[2], 0x06020c0cu);
                            o = (bl & colmask) | (o & ~colmask);
                        }
                        __builtin_amdgcn_raw_buffer_store_b32(o, rdst, offm, y * stride, 0);
                    }
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// Per output slot, everything the descriptor kernel needs to start loading pixels: resolving a slot
// to its level, key and plane takes a chain of dependent small loads, which is cheap with one lane
// per slot here and expensive with a whole lane group waiting on it there.
struct SlotRec {
    long long pix;   // byte offset of the keypoint's pixel inside the pyramid / blur arena
    uint32_t key;    // score << 24 | y << 12 | x (level coordinates)
    uint32_t hi;     // HARRIS_SCORE: ordered response bits
    int level_oi;    // level << 24 | output index
    float scale;     // mvScaleFactor[level]
    float kp_size;
    int valid;
};

__device__ __forceinline__ SlotRec make_slot_rec(const GeomDev* __restrict__ g, const uint32_t* __restrict__ lvl_list, int key_words,
                                                 const int* __restrict__ lvl_count, int f, int slot) {
    SlotRec r;
    r.valid = 0;
    r.pix = 0;
    r.key = r.hi = 0;
    r.level_oi = 0;
    r.scale = r.kp_size = 0.f;
    int level, k, oi;
    uint32_t key, hi;
    if (slot_to_key(g, lvl_list, key_words, lvl_count, f, slot, level, k, key, hi, oi)) {
        const LevelGeom& L = g->lv[level];
        const int x = key & 0xfff, y = (key >> 12) & 0xfff;
        r.pix = (long long)f * g->frame_bytes + L.plane_off + (long long)(y + ORB_EDGE) * L.stride + (x + ORB_EDGE);
        r.key = key;
        r.hi = hi;
        r.level_oi = (level << 24) | oi;
        r.scale = L.scale;
        r.kp_size = L.kp_size;
        r.valid = 1 + L.stride;  // non-zero; carries the plane's row stride
    }
    return r;
}

__global__ __launch_bounds__(256) void k_slots(const GeomDev* __restrict__ g, const uint32_t* __restrict__ lvl_list,
                                               int key_words, const int* __restrict__ lvl_count, SlotRec* __restrict__ recs,
                                               int* __restrict__ n_out) {
    const int f = blockIdx.y;
    const int slot = blockIdx.x * 256 + threadIdx.x;
    if (slot == 0) {
        int n = 0;
        for (int l = 0; l < g->nlevels; l++) n += lvl_count[f * g->nlevels + l];
        n_out[f] = n;
    }
    if (slot >= g->nfeatures) return;
    recs[(long long)f * g->nfeatures + slot] = make_slot_rec(g, lvl_list, key_words, lvl_count, f, slot);
}

// The same records in the order the descriptor kernel should WORK in, which is not the output order: the kernel
// handles eight consecutive records per workgroup, and its time goes into the vector-cache lines the two windows
// of a keypoint touch (every miss of a window line is a trip to the L2).  Neighbouring keypoints' windows overlap, so
// the records of a frame are sorted by (level, position along a Z-order curve over a 16 x 16 grid of the level image):
// a workgroup's eight keypoints then sit next to each other and share lines in the CU's cache.  A record carries its
// output index, so the result does not depend on the order.  One workgroup per frame, counting sort in LDS
// (histogram by atomics, exclusive scan, scatter); frames of up to SLOT_SORT_MAX output slots.
#define SLOT_SORT_THREADS 1024
#define SLOT_SORT_PER 4
#define SLOT_SORT_MAX (SLOT_SORT_THREADS * SLOT_SORT_PER)
#define SLOT_SORT_BINS (ORB_MAX_LEVELS * 256 + 1)  // + one bin behind all others for the empty slots

__device__ __forceinline__ uint32_t morton4(uint32_t x, uint32_t y) {  // 4-bit x, y -> 8-bit interleave
    x = (x | (x << 2)) & 0x33u;
    x = (x | (x << 1)) & 0x55u;
    y = (y | (y << 2)) & 0x33u;
    y = (y | (y << 1)) & 0x55u;
    return x | (y << 1);
}

__global__ __launch_bounds__(SLOT_SORT_THREADS) void k_slots_sorted(const GeomDev* __restrict__ g, const uint32_t* __restrict__ lvl_list,
                                                                    int key_words, const int* __restrict__ lvl_count,
                                                                    SlotRec* __restrict__ recs, int* __restrict__ n_out) {
    __shared__ int hist[SLOT_SORT_BINS];
    __shared__ int wsum[SLOT_SORT_THREADS / 64];
    const int f = blockIdx.x, tid = threadIdx.x, nf = g->nfeatures;
    for (int i = tid; i < SLOT_SORT_BINS; i += SLOT_SORT_THREADS) hist[i] = 0;
    if (tid == 0) {
        int n = 0;
        for (int l = 0; l < g->nlevels; l++) n += lvl_count[f * g->nlevels + l];
        n_out[f] = n;
    }
    __syncthreads();
    SlotRec r[SLOT_SORT_PER];
    int bin[SLOT_SORT_PER], rank[SLOT_SORT_PER];
#pragma unroll
    for (int k = 0; k < SLOT_SORT_PER; k++) {
        const int slot = k * SLOT_SORT_THREADS + tid;
        bin[k] = -1;
        if (slot < nf) {
            r[k] = make_slot_rec(g, lvl_list, key_words, lvl_count, f, slot);
            bin[k] = SLOT_SORT_BINS - 1;
            if (r[k].valid) {
                const int level = r[k].level_oi >> 24;
                const LevelGeom& L = g->lv[level];
                // the level image cut into 16 x 16 cells (cell sides rounded up, so the last cell index is at most 15)
                const uint32_t x = r[k].key & 0xfffu, y = (r[k].key >> 12) & 0xfffu;
                const uint32_t cx = min((x * 16u) / (uint32_t)max(L.w, 1), 15u), cy = min((y * 16u) / (uint32_t)max(L.h, 1), 15u);
                bin[k] = level * 256 + (int)morton4(cx, cy);
            }
            rank[k] = atomicAdd(&hist[bin[k]], 1);
        }
    }
    __syncthreads();
    // exclusive scan of the bins: every thread sums a run of bins, the runs are scanned wave by wave
    const int per = (SLOT_SORT_BINS + SLOT_SORT_THREADS - 1) / SLOT_SORT_THREADS;
    const int b0 = min(tid * per, SLOT_SORT_BINS), b1 = min(b0 + per, SLOT_SORT_BINS);
    int mine = 0;
    for (int b = b0; b < b1; b++) mine += hist[b];
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if ((tid & 63) >= o) incl += v;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int off = incl - mine;
    for (int w = 0; w < (tid >> 6); w++) off += wsum[w];
    for (int b = b0; b < b1; b++) {
        const int c = hist[b];
        hist[b] = off;
        off += c;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SLOT_SORT_PER; k++)
        if (bin[k] >= 0) recs[(long long)f * nf + hist[bin[k]] + rank[k]] = r[k];
}

// IC_Angle (src/ORBextractor.cc:124-151) + computeOrbDescriptor (:154-194) + the output record of
// operator() (:769-777) in one kernel, OD_LPK lanes per keypoint, so that the chain of
// dependent memory round trips (level counts -> key -> pixels) is paid once per keypoint.
//
// Both windows of a keypoint (the 31 disc rows of the UNBLURRED level image, the 37 rows of the BLURRED plane that the
// rotated pattern can reach) are fetched as 16-byte pieces, three per row from the dword at or before the row's first
// byte: 10 rows per load instruction of the keypoint's 32 lanes (30 of them loading), 4 + 4 instructions per lane.
// What a wave-level load costs the texture addresser / vector cache of a CU, measured with tools/ta_probe.hip on
// cache-resident windows: 11 cycles for dwords in rows of 16 lanes, 22 cycles for dwords that are unaligned or lie in
// rows of 11 lanes (37 tag lookups instead of 6 to 10), 23 cycles for 63 lanes x 16 bytes - four times the bytes per
// cycle of the cheapest dword shape, and a fifth of the instructions.
//
// Orientation: intensity-centroid moments over the radius-15 disc.  The lanes of pieces 0 and 1 of a row rebuild the
// row's 8 unaligned dwords (bytes x-15+4q .. x-12+4q) from five aligned ones with v_alignbyte (the fifth comes from the
// next lane); a host-built table gives, per unaligned dword, the byte weights (u + 16 inside the disc, else 0) and the
// inside mask, so that with two v_dot4_u32_u8
//   sum(u * I) = dot4(I, u + 16) - 16 * dot4(I, mask),   sum(v * I) = v * dot4(I, mask).
// The moments are exact integers whatever the summation order.
//
// Descriptor: lane i produces descriptor byte i (8 tests) from the BLURRED plane.  The rotated pattern reaches at most
// 18 px from the keypoint (largest pattern radius 18.38), so the 37 x 37 window around it goes to LDS (37 rows of three
// 16-byte pieces) and the 16 scattered samples of each lane are LDS byte reads; gathering them from global memory
// instead costs one vector-cache tag lookup per lane and sample.
#define DS_REACH 18
#define DS_ROWS (2 * DS_REACH + 1)
#define DS_PIECES 3                 // 16-byte pieces per window row: 48 bytes >= 37 + up to 3 bytes of lead-in
#define DS_DWORDS (4 * DS_PIECES)   // LDS pitch of a window row in dwords
// Lanes per keypoint.  Every lane of a keypoint computes the same fastAtan2 / cosf / sinf, so fewer
// lanes per keypoint mean less redundant arithmetic per wave; 32 lanes produce one descriptor byte each.
#define OD_LPK 32
#define OD_KPB (256 / OD_LPK)  // keypoints per workgroup and round
#define OD_ROUNDS 1
#define OD_BYTES (32 / OD_LPK)  // descriptor bytes per lane
#define OD_ROWS_PER_IT (OD_LPK / DS_PIECES)  // window rows per load instruction of a keypoint's lanes
#define DS_ITERS ((DS_ROWS + OD_ROWS_PER_IT - 1) / OD_ROWS_PER_IT)
#define OR_ITERS ((31 + OD_ROWS_PER_IT - 1) / OD_ROWS_PER_IT)
typedef uint32_t od_u32x4 __attribute__((ext_vector_type(4)));

// Developer instrumentation (-DORBHIP_OD_PHASE_TIMING, never in a shipped build): shader-clock cycles the waves of
// k_orient_describe spend in each phase, summed; read and reset through orbhip_debug_od_phase.
#ifdef ORBHIP_OD_PHASE_TIMING
__device__ unsigned long long od_phase[8];
#define OD_T0 unsigned long long odt_ = __builtin_readcyclecounter();
#define OD_T(k)                                                                           \
    {                                                                                     \
        const unsigned long long now_ = __builtin_readcyclecounter();                     \
        if ((threadIdx.x & 63) == 0) atomicAdd(&od_phase[k], now_ - odt_);                \
        odt_ = now_;                                                                      \
    }
#else
#define OD_T0
#define OD_T(k)
#endif

__global__ __launch_bounds__(256) void k_orient_describe(const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                                         const GeomDev* __restrict__ g,
                                                         const SlotRec* __restrict__ recs, int key_words,
                                                         const uint32_t* __restrict__ otab, float* __restrict__ angles,
                                                         const int8_t* __restrict__ pattern,
                                                         OrbKeypointDev* __restrict__ kp_out, uint8_t* __restrict__ desc_out,
                                                         int cap) {
    __shared__ __attribute__((aligned(16))) uint32_t ds_win[OD_KPB][DS_ROWS * DS_DWORDS];
    // the disc weight table and the test pattern are the same for every keypoint: one copy per
    // workgroup in LDS instead of a trip through the vector cache per keypoint
    __shared__ uint2 s_otab[31 * 8];
    // the 256 tests as floats (x0, y0, x1, y1), test 8 i + j of lane i at [32 j + i]: one conflict-free 16-byte read
    // per test instead of byte reads and four conversions
    __shared__ float4 s_patf[256];
    OD_T0
    // The tables go to LDS through registers, and the barrier that publishes them comes only after the slot record
    // and both windows have been requested: the three memory round trips of a workgroup (tables, record, windows)
    // would otherwise run one after the other, and a workgroup lives for little else.
    uint2 t_otab = make_uint2(0u, 0u);
    if (threadIdx.x < 31 * 8) t_otab = ((const uint2*)otab)[threadIdx.x];
    const uint32_t t_pat = ((const uint32_t*)pattern)[threadIdx.x];  // test threadIdx.x: four signed bytes
    int bx = blockIdx.x, f = blockIdx.y;
    xcd_frame_swizzle(bx, f, (int)gridDim.x, (int)gridDim.y);
    const int li = threadIdx.x % OD_LPK;
    OD_T(0)
    const int slot = bx * OD_KPB + (threadIdx.x / OD_LPK);
    SlotRec rec;
    rec.valid = 0;
    rec.pix = 0;
    rec.key = rec.hi = 0;
    rec.level_oi = 0;
    rec.scale = rec.kp_size = 0.f;
    if (slot < g->nfeatures) rec = recs[(long long)f * g->nfeatures + slot];
    const bool active = rec.valid != 0;
    const uint32_t key = rec.key, hi = rec.hi;
    const int level = rec.level_oi >> 24, oi = rec.level_oi & 0xffffff;
    const int x = key & 0xfff, y = (key >> 12) & 0xfff;
    const int stride = rec.valid - 1;
    const long long pix = rec.pix;
    OD_T(1)

    // all loads of both windows are issued before anything waits on them.  Buffer loads over the frame's two
    // arenas (the frame is the same for the whole workgroup): 32-bit offsets, no per-lane 64-bit address arithmetic
    const long long fbase = (long long)f * g->frame_bytes;
    const __amdgpu_buffer_rsrc_t rs_pyr = plane_rsrc(pyr + fbase, (int)g->frame_bytes),
                                 rs_blur = plane_rsrc(blur + fbase, (int)g->frame_bytes);
    const uint32_t pixf = (uint32_t)(pix - fbase);
    // rows start at the dword at or before their first byte (planes and strides are 4-byte aligned)
    const int lead = (int)((pixf - DS_REACH) & 3u), olead = (int)((pixf - ORB_HALF_PATCH) & 3u);
    const uint32_t ocorner = pixf - (uint32_t)(ORB_HALF_PATCH * stride + ORB_HALF_PATCH + olead);
    const uint32_t corner = pixf - (uint32_t)(DS_REACH * stride + DS_REACH + lead);
    const int rr = li / DS_PIECES, pc = li - DS_PIECES * rr;  // row within the group of rows of one instruction, piece
    const bool loader = li < OD_ROWS_PER_IT * DS_PIECES;
    const uint32_t lane_off = __umul24((uint32_t)rr, (uint32_t)stride) + (uint32_t)(16 * pc);
    od_u32x4 disc[OR_ITERS], wreg[DS_ITERS];
#pragma unroll
    for (int it = 0; it < OR_ITERS; it++) disc[it] = (od_u32x4)(0u);
#pragma unroll
    for (int it = 0; it < DS_ITERS; it++) wreg[it] = (od_u32x4)(0u);
    if (active && loader) {
#pragma unroll
        for (int it = 0; it < OR_ITERS; it++)
            if (it * OD_ROWS_PER_IT + rr < 31)
                disc[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_pyr, ocorner + lane_off, it * OD_ROWS_PER_IT * stride, 0);
#pragma unroll
        for (int it = 0; it < DS_ITERS; it++)
            if (it * OD_ROWS_PER_IT + rr < DS_ROWS)
                wreg[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_blur, corner + lane_off, it * OD_ROWS_PER_IT * stride, 0);
    }
    if (threadIdx.x < 31 * 8) s_otab[threadIdx.x] = t_otab;
    s_patf[32 * (threadIdx.x & 7) + (threadIdx.x >> 3)] =
        make_float4((float)(int8_t)(t_pat & 0xffu), (float)(int8_t)((t_pat >> 8) & 0xffu), (float)(int8_t)((t_pat >> 16) & 0xffu),
                    (float)(int8_t)(t_pat >> 24));
    __syncthreads();
    if (!active) return;

    OD_T(2)
    int m10 = 0, m01 = 0;
#pragma unroll
    for (int it = 0; it < OR_ITERS; it++) {
        const int row = it * OD_ROWS_PER_IT + rr;
        // the aligned dword after this lane's four: the first of the next piece of the row (every lane shuffles; the
        // lanes of the last piece read a lane they do not need)
        const uint32_t d4 = (uint32_t)__shfl_down((int)disc[it].x, 1);
        if (loader && pc < 2 && row < 31) {
            const int v = row - ORB_HALF_PATCH;
            const uint32_t d[5] = {disc[it].x, disc[it].y, disc[it].z, disc[it].w, d4};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t px = __builtin_amdgcn_alignbyte(d[q + 1], d[q], (uint32_t)olead);
                const uint2 w = s_otab[row * 8 + 4 * pc + q];
                const int sm = (int)__builtin_amdgcn_udot4(px, w.y, 0u, false);
                const int tm = (int)__builtin_amdgcn_udot4(px, w.x, 0u, false);
                m10 += tm - 16 * sm;
                m01 += v * sm;
            }
        }
    }
#pragma unroll
    for (int o = OD_LPK / 2; o > 0; o >>= 1) {
        m10 += __shfl_xor(m10, o);
        m01 += __shfl_xor(m01, o);
    }
    const float angle_deg = fast_atan2_deg((float)m01, (float)m10);
    if (li == 0) angles[(long long)f * g->nfeatures + slot] = angle_deg;

    OD_T(3)
    uint32_t* win = ds_win[threadIdx.x / OD_LPK];
#pragma unroll
    for (int it = 0; it < DS_ITERS; it++) {
        const int row = it * OD_ROWS_PER_IT + rr;
        if (loader && row < DS_ROWS) *(od_u32x4*)(win + row * DS_DWORDS + 4 * pc) = wreg[it];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the window is private to this keypoint's lanes

    OD_T(4)
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float angle = angle_deg * factorPI;
    const float a = orbtrig::cos_f<false>(angle), b = orbtrig::sin_f<false>(angle);
    // cvRound of the rotated pattern coordinates without a conversion: t + 1.5 * 2^23 rounds t (|t| < 2^22) to the
    // nearest integer, ties to even, exactly like cvRound's rint, and leaves 0x4B400000 + rint(t) in the float's bits;
    // the low 24 bits (0x400000 + rint(t)) feed a 24-bit multiply-add, and the constants that ride along,
    // 0x400000 * pitch + 0x4B400000, come off the window's base address once.
    const float magic = 12582912.f;
    const uint32_t pitch = 4 * DS_DWORDS;
    const uint8_t* center = (const uint8_t*)win + DS_REACH * (4 * DS_DWORDS) + DS_REACH + lead;
    const uint8_t* cbase = center - (0x400000u * pitch + 0x4B400000u);
    static_assert(OD_BYTES == 1 && OD_LPK == 32, "the float pattern table is laid out for 8 tests per lane, 32 lanes per keypoint");
    uint32_t val = 0;
#pragma unroll
    for (int j = 0; j < 8 * OD_BYTES; j++) {
        const float4 pt = s_patf[32 * j + li];
        const float x0 = pt.x, y0 = pt.y, x1 = pt.z, y1 = pt.w;
        const uint32_t r0 = __float_as_uint(__fadd_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)), magic));
        const uint32_t c0 = __float_as_uint(__fadd_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)), magic));
        const uint32_t r1 = __float_as_uint(__fadd_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)), magic));
        const uint32_t c1 = __float_as_uint(__fadd_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)), magic));
        const int v0 = cbase[__umul24(r0, pitch) + c0], v1 = cbase[__umul24(r1, pitch) + c1];
        val |= (uint32_t)(v0 < v1) << j;
    }
    OD_T(5)
    if (oi >= cap) return;
    const long long o = (long long)f * cap + oi;
    if (OD_BYTES == 1) desc_out[o * 32 + li] = (uint8_t)val;
    else if (OD_BYTES == 2) ((uint16_t*)(desc_out + o * 32))[li] = (uint16_t)val;
    else ((uint32_t*)(desc_out + o * 32))[li] = val;
    if (li < 7) {
        uint32_t w;
        const float scale = rec.scale;
        switch (li) {
            case 0: w = __float_as_uint(level ? __fmul_rn((float)x, scale) : (float)x); break;
            case 1: w = __float_as_uint(level ? __fmul_rn((float)y, scale) : (float)y); break;
            case 2: w = __float_as_uint(rec.kp_size); break;
            case 3: w = __float_as_uint(angle_deg); break;
            case 4: w = key_words == 2 ? __float_as_uint(ordered_to_float(hi)) : __float_as_uint((float)(key >> 24)); break;
            case 5: w = (uint32_t)level; break;
            default: w = 0xffffffffu; break;
        }
        ((uint32_t*)(kp_out + o))[li] = w;
    }
    OD_T(6)
}

// ---------------------------------------------------------------------------------------------
// Device self-tests of the arithmetic helpers (test instrumentation).
__global__ void k_test_atan2(const float* yx, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fast_atan2_deg(yx[2 * i], yx[2 * i + 1]);
}
__global__ void k_test_sincos(const float* in, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        out[2 * i] = orbtrig::cos_f<false>(in[i]);
        out[2 * i + 1] = orbtrig::sin_f<false>(in[i]);
    }
}
__global__ void k_test_select(uint32_t* keys, int n, int nth) {
    if (blockIdx.x == 0 && threadIdx.x == 0) orbsel::nth_element_desc(keys, n, nth);
}
__global__ void k_test_wave_select(uint32_t* keys, int n, int nth, int depth) {
    __shared__ uint32_t v[4096];
    __shared__ uint16_t lp[4096], rp[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < n; i += 64) v[i] = keys[i];
    orbsel::wave_lds_fence();
    orbsel::wave_nth_element_desc(v, n, nth, lp, rp, lane, depth);
    for (int i = lane; i < n; i += 64) keys[i] = v[i];
}
__global__ void k_test_measure(const uint8_t* patches, int* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fast_measure(patches + (long long)i * 49 + 24, 7);
}

}  // namespace orbk

#ifdef ORBHIP_RZ_PHASE_TIMING
extern "C" int orbhip_debug_rz_phase(unsigned long long* out8) {
    unsigned long long z[8] = {0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(orbk::rz_phase), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(orbk::rz_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

#ifdef ORBHIP_OD_PHASE_TIMING
extern "C" int orbhip_debug_od_phase(unsigned long long* out8) {
    unsigned long long z[8] = {0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(orbk::od_phase), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(orbk::od_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

// ---------------------------------------------------------------------------------------------
// Launchers (host side of this translation unit).
namespace orbk {

void launch_level0(hipStream_t st, const GeomDev& hg, int batch, const uint8_t* imgs, long long frame_stride,
                   long long row_stride, uint8_t* pyr, const GeomDev* dg) {
    const LevelGeom& L = hg.lv[0];
    const int wide = ((uintptr_t)imgs % 16 == 0) && (row_stride % 16 == 0) && (frame_stride % 16 == 0);
    const int gpr = L.stride / 16;  // >= 2 (a row holds at least the two 16-pixel borders)
    const uint32_t gpr_inv = (uint32_t)(0x100000000ULL / (uint32_t)gpr) + 1u;
    dim3 grid(((unsigned)gpr * (unsigned)L.ph + 255u) / 256u, batch), block(256);
    hipLaunchKernelGGL(k_level0, grid, block, 0, st, imgs, frame_stride, row_stride, pyr, dg, wide, gpr_inv);
}

void launch_resize(hipStream_t st, const GeomDev& hg, int batch, int level, uint8_t* pyr, const GeomDev* dg,
                   const int* xofs, const uint32_t* xab, const uint32_t* ysrc, const uint32_t* yab) {
    const LevelGeom& L = hg.lv[level];
    const LevelGeom& P = hg.lv[level - 1];
    // source rectangle of a 256 x RZ_TILE_H destination tile at this level's scale, plus alignment slack
    const double sx = (double)P.w / L.w, sy = (double)P.h / L.h;
    int lds_pitch = ((int)(256 * sx) + 3 + 15 + 15) & ~15;
    int lds_rows = (int)(RZ_TILE_H * sy) + 4;
    if ((size_t)lds_pitch * lds_rows > 60 * 1024) lds_rows = (60 * 1024) / lds_pitch;
    dim3 grid((L.stride / 4 + 63) / 64, (L.ph + RZ_TILE_H - 1) / RZ_TILE_H, batch), block(64, 4);
    hipLaunchKernelGGL(k_resize, grid, block, (size_t)lds_pitch * lds_rows, st, pyr, dg, level, xofs, xab, ysrc, yab,
                       lds_pitch, lds_rows);
}

void launch_score(hipStream_t st, const GeomDev& hg, int batch, int level, const uint8_t* pyr, uint8_t* score,
                  const GeomDev* dg) {
    const LevelGeom& L = hg.lv[level];
    const int iw = L.w - 2 * ORB_EDGE, ih = L.h - 2 * ORB_EDGE;
    if (iw <= 0 || ih <= 0) return;
    dim3 grid((iw + 63) / 64, (ih + 3) / 4, batch), block(64, 4);
    hipLaunchKernelGGL(k_score, grid, block, 0, st, pyr, score, dg, level);
}

void launch_nms(hipStream_t st, const GeomDev& hg, int batch, int max_cell_h, const uint8_t* score, const GeomDev* dg,
                const CellDesc* cells, uint32_t* cand, int* n_total, int* err) {
    dim3 grid(hg.cells_per_frame, batch), block(256);
    const size_t lds = (size_t)(2 * max_cell_h + 516) * sizeof(int);
    hipLaunchKernelGGL(k_nms, grid, block, lds, st, score, dg, cells, cand, n_total, err);
}

// keys of the largest level list, rounded up to 256 and capped at SEL_LCAP
static int level_select_cap(const GeomDev& hg) {
    int cap = 256;
    for (int l = 0; l < hg.nlevels; l++) cap = std::max(cap, hg.lv[l].lvl_cap);
    return std::min((cap + 255) & ~255, SEL_LCAP);
}

void launch_select(hipStream_t st, const GeomDev& hg, int batch, const GeomDev* dg, const CellDesc* cells,
                   uint32_t* cand, const int* n_total, int* n_retain, uint32_t* lvl_list, int* lvl_count, int* err) {
    dim3 per_level(hg.nlevels, batch), per_cell((hg.cells_per_frame + 3) / 4, batch), block(256);
    hipLaunchKernelGGL(k_quota, per_level, block, 0, st, dg, cells, n_total, n_retain);
    hipLaunchKernelGGL(k_cell_select<uint32_t>, per_cell, block, 0, st, dg, cells, cand, n_total, n_retain);
    const int lcap = level_select_cap(hg);
    hipLaunchKernelGGL(k_level_select<uint32_t>, per_level, block, (size_t)lcap * (4 + 4), st, dg, cells, cand, n_retain,
                       lvl_list, lvl_count, err, lcap);
}

void launch_harris_select(hipStream_t st, const GeomDev& hg, int batch, const uint8_t* pyr, const GeomDev* dg,
                          const CellDesc* cells, const uint32_t* cand, uint64_t* cand64, const int* n_total, int* n_retain,
                          uint64_t* lvl_list64, int* lvl_count, int* err) {
    dim3 per_level(hg.nlevels, batch), per_cell4((hg.cells_per_frame + 3) / 4, batch), per_cell(hg.cells_per_frame, batch),
        block(256);
    hipLaunchKernelGGL(k_harris, per_cell, block, 0, st, pyr, dg, cells, cand, n_total, cand64);
    hipLaunchKernelGGL(k_quota, per_level, block, 0, st, dg, cells, n_total, n_retain);
    hipLaunchKernelGGL(k_cell_select<uint64_t>, per_cell4, block, 0, st, dg, cells, cand64, n_total, n_retain);
    const int lcap = level_select_cap(hg);
    hipLaunchKernelGGL(k_level_select<uint64_t>, per_level, block, (size_t)lcap * (8 + 4), st, dg, cells, cand64, n_retain,
                       lvl_list64, lvl_count, err, lcap);
}

void launch_blur(hipStream_t st, const GeomDev& hg, int batch, const uint8_t* pyr, uint8_t* blur, const GeomDev* dg,
                 const int taps[7], bool tie_even) {
    BlurPlan plan;
    plan.nlevels = hg.nlevels;
    int total = 0;
    for (int l = 0; l < hg.nlevels; l++) {
        const LevelGeom& L = hg.lv[l];
        plan.first[l] = total;
        plan.nx[l] = (L.stride / 4 + 63) / 64;
        total += plan.nx[l] * ((L.ph + 4 * BL_ROWS - 1) / (4 * BL_ROWS));
    }
    plan.first[hg.nlevels] = total;
    dim3 grid(total, batch), block(64, 4);
    if (tie_even)
        hipLaunchKernelGGL(k_blur<true>, grid, block, 0, st, pyr, blur, dg, plan, taps[0], taps[1], taps[2], taps[3]);
    else
        hipLaunchKernelGGL(k_blur<false>, grid, block, 0, st, pyr, blur, dg, plan, taps[0], taps[1], taps[2], taps[3]);
}

void launch_orient_describe(hipStream_t st, const GeomDev& hg, int batch, const uint8_t* pyr, const uint8_t* blur,
                            const GeomDev* dg, const uint32_t* lvl_list, int key_words, const int* lvl_count,
                            const uint32_t* otab, float* angles, const int8_t* pattern, void* slot_recs,
                            OrbKeypointDev* kp_out, uint8_t* desc_out, int* n_out, int cap, hipEvent_t blur_done) {
    static const bool sorted = [] {
        const char* e = getenv("ORBHIP_SLOT_SORT");  // 0: records in output order (tuning experiments)
        return !(e && e[0] == '0');
    }();
    if (sorted && hg.nfeatures <= SLOT_SORT_MAX)
        hipLaunchKernelGGL(k_slots_sorted, dim3(batch), dim3(SLOT_SORT_THREADS), 0, st, dg, lvl_list, key_words, lvl_count,
                           (SlotRec*)slot_recs, n_out);
    else
        hipLaunchKernelGGL(k_slots, dim3((hg.nfeatures + 255) / 256, batch), dim3(256), 0, st, dg, lvl_list, key_words, lvl_count,
                           (SlotRec*)slot_recs, n_out);
    // the blurred plane may come from another stream (blur_done): only the descriptor kernel needs it
    if (blur_done) (void)hipStreamWaitEvent(st, blur_done, 0);
    dim3 grid((hg.nfeatures + OD_KPB * OD_ROUNDS - 1) / (OD_KPB * OD_ROUNDS), batch), block(256);
    hipLaunchKernelGGL(k_orient_describe, grid, block, 0, st, pyr, blur, dg, (const SlotRec*)slot_recs, key_words, otab, angles,
                       pattern, kp_out, desc_out, cap);
}

size_t slot_record_bytes() { return sizeof(SlotRec); }

void launch_test_atan2(hipStream_t st, const float* yx, float* out, int n) {
    hipLaunchKernelGGL(k_test_atan2, dim3((n + 255) / 256), dim3(256), 0, st, yx, out, n);
}
void launch_test_sincos(hipStream_t st, const float* in, float* out, int n) {
    hipLaunchKernelGGL(k_test_sincos, dim3((n + 255) / 256), dim3(256), 0, st, in, out, n);
}
void launch_test_select(hipStream_t st, uint32_t* keys, int n, int nth) {
    hipLaunchKernelGGL(k_test_select, dim3(1), dim3(64), 0, st, keys, n, nth);
}
void launch_test_wave_select(hipStream_t st, uint32_t* keys, int n, int nth, int depth) {
    hipLaunchKernelGGL(k_test_wave_select, dim3(1), dim3(64), 0, st, keys, n, nth, depth);
}
void launch_test_measure(hipStream_t st, const uint8_t* patches, int* out, int n) {
    hipLaunchKernelGGL(k_test_measure, dim3((n + 255) / 256), dim3(256), 0, st, patches, out, n);
}

}  // namespace orbk
