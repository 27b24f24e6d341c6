// Fused per-cell FAST for gfx950: one workgroup per (cell, frame) does what the reference does per
// cell in src/ORBextractor.cc:599-614 — cv::FAST(cellImage, kps, fastTh, true), and again with
// threshold 7 when that found <= 3 keypoints — entirely out of LDS:
//
//   1. the cell image (examined pixels + 3-px halo) is staged into LDS with aligned 16-byte loads;
//   2. QUICK TEST: every examined pixel is tested on the four compass pixels of the Bresenham
//      circle (a 9-arc always contains one of {0,8} and one of {4,12}), a necessary condition for
//      a corner at the lower of the two thresholds, two pixels per packed 16-bit instruction;
//      survivors are compacted into a per-wave list with wave ballots;
//   3. MEASURE: the exact corner measure M = max(A, B) (fast_measure.h) is computed only for listed
//      pixels (dense lanes), written to an LDS score tile that is zero elsewhere, and pixels with
//      M above the lower threshold are appended, still in raster order, to a per-wave corner list;
//   4. NMS + ORDERED OUTPUT over the corner list only: a pixel is a keypoint at threshold t iff
//      M > t and M is a strict maximum of its 8 neighbours inside the cell (non-corners and
//      out-of-cell pixels count as 0, which a strict maximum of the raw measure reproduces because
//      M > t >= any non-corner's measure).  Each wave owns a contiguous band of rows, so per-wave
//      counts, a prefix over the four waves and ballot/mbcnt prefix sums give raster order.
//      If a wave's corner list overflows (very dense texture) the block falls back to scanning
//      the score tile densely.
//
// Pyramid bytes are read from HBM once; no score plane is written.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "extract_kernels.h"
#include "fast_measure.h"

namespace orbk {

#define FC_ROWS_PER_ITER 4

typedef short short2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t byte_of(uint32_t w, int j) { return (w >> (8 * j)) & 0xffu; }
__device__ __forceinline__ int lane_prefix(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}
__device__ __forceinline__ short2v as_s2(uint32_t v) { return __builtin_bit_cast(short2v, v); }
__device__ __forceinline__ uint32_t as_u32(short2v v) { return __builtin_bit_cast(uint32_t, v); }
// (byte lo | byte hi << 16) of two source bytes; selector bytes 0-3 pick from b, 4-7 from a, 0x0c = 0
#define PAIR(a, b, sel) as_s2(__builtin_amdgcn_perm((a), (b), (sel)))

struct FcLayout {
    int pitch;     // bytes per image-tile row (multiple of 16)
    int spitch;    // bytes per score-tile row (multiple of 4)
    int img_bytes, sc_bytes;
};

// maximum of the 8 neighbours of the score byte at p (score tile has a zero border)
__device__ __forceinline__ uint32_t nb_max8(const uint8_t* __restrict__ p, int sp) {
    const uint32_t a = max(max((uint32_t)p[-sp - 1], (uint32_t)p[-sp]), (uint32_t)p[-sp + 1]);
    const uint32_t b = max(max((uint32_t)p[sp - 1], (uint32_t)p[sp]), (uint32_t)p[sp + 1]);
    return max(max(a, b), max((uint32_t)p[-1], (uint32_t)p[1]));
}

__global__ __launch_bounds__(256) void k_fast_cell(const uint8_t* __restrict__ pyr, const GeomDev* __restrict__ g,
                                                   const CellDesc* __restrict__ cells, uint32_t* __restrict__ cand,
                                                   int* __restrict__ n_total, int* __restrict__ err, int qcap,
                                                   int ccap) {
    extern __shared__ uint4 lds4[];
    __shared__ int wtot[4][2];
    __shared__ int overflow;
    const int c = blockIdx.x, f = blockIdx.y;
    const CellDesc cd = cells[c];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int* out_total = n_total + (long long)f * g->cells_per_frame + c;
    if (cd.cw <= 0 || cd.ch <= 0) {
        if (tid == 0) *out_total = 0;
        return;
    }
    const LevelGeom& L = g->lv[cd.level];
    const int stride = L.stride;
    const uint8_t* plane = pyr + (long long)f * g->frame_bytes + L.plane_off;
    const int cw = cd.cw, ch = cd.ch;
    const int gx0 = ORB_EDGE + cd.x0 - 3;  // padded-plane column of the tile's first needed byte
    const int gxs = gx0 & ~15;
    const int shift = gx0 - gxs;
    const int th = ch + 6;
    const int pitch = (shift + cw + 6 + 15) & ~15;
    const int spitch = (cw + 2 + 3) & ~3;
    const int gy0 = ORB_EDGE + cd.y0 - 3;
    uint8_t* imgb = (uint8_t*)lds4;
    uint8_t* scb = imgb + pitch * th;                 // (ch + 2) rows of spitch bytes, zero border
    uint16_t* qlist = (uint16_t*)(scb + ((spitch * (ch + 2) + 15) & ~15)) + wave * qcap;
    uint32_t* clist = (uint32_t*)((uint16_t*)(scb + ((spitch * (ch + 2) + 15) & ~15)) + 4 * qcap) + wave * ccap;

    {
        const int pc = pitch >> 4;
        const int nchunk = pc * th;
        uint4* dst = (uint4*)imgb;
        for (int i0 = 0; i0 < nchunk; i0 += 1024) {
            // four 16-byte loads in flight per lane before the LDS stores
            const int ia = min(i0 + tid, nchunk - 1), ib = min(i0 + 256 + tid, nchunk - 1);
            const int ic = min(i0 + 512 + tid, nchunk - 1), id = min(i0 + 768 + tid, nchunk - 1);
            const int ra = ia / pc, rb2 = ib / pc, rc = ic / pc, rd = id / pc;
            const uint4 va = *(const uint4*)(plane + (long long)(gy0 + ra) * stride + gxs + 16 * (ia - ra * pc));
            const uint4 vb = *(const uint4*)(plane + (long long)(gy0 + rb2) * stride + gxs + 16 * (ib - rb2 * pc));
            const uint4 vc = *(const uint4*)(plane + (long long)(gy0 + rc) * stride + gxs + 16 * (ic - rc * pc));
            const uint4 vd = *(const uint4*)(plane + (long long)(gy0 + rd) * stride + gxs + 16 * (id - rd * pc));
            dst[ia] = va;
            dst[ib] = vb;
            dst[ic] = vc;
            dst[id] = vd;
        }
        uint32_t* s32 = (uint32_t*)scb;
        for (int i = tid; i < (spitch * (ch + 2)) >> 2; i += 256) s32[i] = 0;
        if (tid == 0) overflow = 0;
    }
    __syncthreads();

    const int tHi = min(max(g->fast_th, 0), 255), tLo = 7;
    const int tmin = min(tHi, tLo);
    const int c_lo = shift + 3;  // examined tile columns [c_lo, c_lo + cw)
    const int g_first = c_lo >> 2;
    const int gg = g_first + lane;  // this lane's 4-pixel group (tile columns 4gg .. 4gg+3)
    const bool act = 4 * gg < c_lo + cw;
    // lanes whose pixel j of the group is an examined pixel
    const unsigned long long vm0 = __ballot(act && 4 * gg + 0 >= c_lo && 4 * gg + 0 < c_lo + cw);
    const unsigned long long vm1 = __ballot(act && 4 * gg + 1 >= c_lo && 4 * gg + 1 < c_lo + cw);
    const unsigned long long vm2 = __ballot(act && 4 * gg + 2 >= c_lo && 4 * gg + 2 < c_lo + cw);
    const unsigned long long vm3 = __ballot(act && 4 * gg + 3 >= c_lo && 4 * gg + 3 < c_lo + cw);
    const int rpw = (ch + 3) >> 2;
    const int r_begin = min(wave * rpw, ch), r_end = min(r_begin + rpw, ch);
    const short2v thr1 = {(short)(tmin + 1), (short)(tmin + 1)};
    const short2v nthr1 = {(short)(-tmin - 1), (short)(-tmin - 1)};
    int nc = 0;  // corners (M > tmin) of this wave's band, raster order

    // ---- quick test + exact measure ------------------------------------------------------------
    for (int rb = r_begin; rb < r_end; rb += FC_ROWS_PER_ITER) {
        int n = 0;
        const int rend = min(rb + FC_ROWS_PER_ITER, r_end);
        for (int r = rb; r < rend; r++) {
            uint32_t z0 = 0xffffffffu, z1 = 0xffffffffu;
            if (act) {
                const uint32_t* row = (const uint32_t*)(imgb + (r + 3) * pitch) + gg;
                const uint32_t wl = row[-1], wm = row[0], wr = row[1];
                const uint32_t up = *(const uint32_t*)((const uint8_t*)row - 3 * pitch);
                const uint32_t dn = *(const uint32_t*)((const uint8_t*)row + 3 * pitch);
                // pixels (0,1) and (2,3) of the group, two per packed op
                const short2v V0 = PAIR(wm, wm, 0x0c010c00u), V1 = PAIR(wm, wm, 0x0c030c02u);
                const short2v U0 = PAIR(up, up, 0x0c010c00u), U1 = PAIR(up, up, 0x0c030c02u);
                const short2v D0 = PAIR(dn, dn, 0x0c010c00u), D1 = PAIR(dn, dn, 0x0c030c02u);
                const short2v E0 = PAIR(wr, wm, 0x0c040c03u), E1 = PAIR(wr, wr, 0x0c020c01u);
                const short2v W0 = PAIR(wl, wl, 0x0c020c01u), W1 = PAIR(wm, wl, 0x0c040c03u);
#define QTEST(V, U, D, E, W, z)                                                                                  \
    {                                                                                                            \
        const short2v d0 = V - D, d8 = V - U, d4 = V - E, d12 = V - W;                                           \
        const short2v a = __builtin_elementwise_min(__builtin_elementwise_max(d0, d8), __builtin_elementwise_max(d4, d12)); \
        const short2v b = __builtin_elementwise_max(__builtin_elementwise_min(d0, d8), __builtin_elementwise_min(d4, d12)); \
        /* pass <=> a > tmin || b < -tmin <=> (a - (tmin+1)) >= 0 || ((-tmin-1) - b) >= 0 */                     \
        z = as_u32(a - thr1) & as_u32(nthr1 - b);                                                                \
    }
                QTEST(V0, U0, D0, E0, W0, z0);
                QTEST(V1, U1, D1, E1, W1, z1);
#undef QTEST
            }
            // raster order inside the row: a lane's pixels follow those of all lower lanes
            const bool f0 = (z0 & 0x8000u) == 0, f1 = (z0 & 0x80000000u) == 0;
            const bool f2 = (z1 & 0x8000u) == 0, f3 = (z1 & 0x80000000u) == 0;
            const unsigned long long m0 = __ballot(f0) & vm0, m1 = __ballot(f1) & vm1;
            const unsigned long long m2 = __ballot(f2) & vm2, m3 = __ballot(f3) & vm3;
            int pos = n + lane_prefix(m0) + lane_prefix(m1) + lane_prefix(m2) + lane_prefix(m3);
            const uint32_t ebase = (uint32_t)(r << 8) | (uint32_t)(4 * gg);
            if ((m0 >> lane) & 1) qlist[pos++] = (uint16_t)ebase;
            if ((m1 >> lane) & 1) qlist[pos++] = (uint16_t)(ebase + 1);
            if ((m2 >> lane) & 1) qlist[pos++] = (uint16_t)(ebase + 2);
            if ((m3 >> lane) & 1) qlist[pos++] = (uint16_t)(ebase + 3);
            n += __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
        }
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            uint32_t e = 0;
            int m = 0;
            if (i < n) {
                e = qlist[i];
                m = min(fast_measure(imgb + ((e >> 8) + 3) * pitch + (e & 255), pitch), 255);
            }
            const int r = e >> 8, cc = e & 255;
            const bool corner = m > tmin;
            if (corner) scb[(r + 1) * spitch + (cc - c_lo + 1)] = (uint8_t)m;
            const unsigned long long mask = __ballot(corner);
            if (corner) {
                const int pos = nc + lane_prefix(mask);
                if (pos < ccap) clist[pos] = (uint32_t)m << 16 | e;
            }
            nc += __popcll(mask);
        }
    }
    if (nc > ccap && lane == 0) overflow = 1;
    __syncthreads();

    // ---- strict-maximum NMS, adaptive threshold, raster-ordered output ---------------------------
    uint32_t* out = cand + (long long)f * g->cand_per_frame + cd.cand_off;
    if (!overflow) {
        int nHi = 0, nLo = 0;
        for (int i = lane; i < nc; i += 64) {
            uint32_t e = clist[i];
            const int r = (e >> 8) & 255, cc = e & 255;
            const uint32_t m = e >> 16;
            const bool lm = m > nb_max8(scb + (r + 1) * spitch + (cc - c_lo + 1), spitch);
            const bool fHi = lm && (int)m > tHi, fLo = lm && (int)m > tLo;
            clist[i] = e | (fHi ? 0x1000000u : 0u) | (fLo ? 0x2000000u : 0u);
            nHi += fHi;
            nLo += fLo;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            nHi += __shfl_xor(nHi, o);
            nLo += __shfl_xor(nLo, o);
        }
        if (lane == 0) {
            wtot[wave][0] = nHi;
            wtot[wave][1] = nLo;
        }
        __syncthreads();
        const int totHi = wtot[0][0] + wtot[1][0] + wtot[2][0] + wtot[3][0];
        const bool useLo = totHi <= 3;  // "if(cellKeyPoints[i][j].size()<=3)" -> FAST(cellImage, ..., 7, true)
        const int sel = useLo ? 1 : 0;
        const uint32_t fbit = useLo ? 0x2000000u : 0x1000000u;
        int off = 0;
        for (int w2 = 0; w2 < wave; w2++) off += wtot[w2][sel];
        int total = wtot[0][sel] + wtot[1][sel] + wtot[2][sel] + wtot[3][sel];
        if (total > cd.cand_cap) {
            if (tid == 0) atomicOr(err, 1);
            total = cd.cand_cap;
        }
        for (int i0 = 0; i0 < nc; i0 += 64) {
            const int i = i0 + lane;
            uint32_t e = 0;
            if (i < nc) e = clist[i];
            const bool fl = (e & fbit) != 0;
            const unsigned long long mask = __ballot(fl);
            if (fl) {
                const int pos = off + lane_prefix(mask);
                if (pos < cd.cand_cap) {
                    const uint32_t r = (e >> 8) & 255, cc = e & 255, s = ((e >> 16) & 255) - 1u;
                    // cell-image coordinates: x - iniX = tile column - shift, y - iniY = r + 3
                    out[pos] = (s << 24) | ((r + 3) << 12) | (cc - shift);
                }
            }
            off += __popcll(mask);
        }
        if (tid == 0) *out_total = total;
        return;
    }

    // dense fallback: scan the wave's band of the score tile
    int nHi = 0, nLo = 0;
    for (int r = r_begin; r < r_end; r++)
        for (int x = lane; x < cw; x += 64) {
            const uint8_t* p = scb + (r + 1) * spitch + x + 1;
            const uint32_t m = p[0];
            if (m != 0 && m > nb_max8(p, spitch)) {
                nHi += (int)m > tHi;
                nLo += (int)m > tLo;
            }
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        nHi += __shfl_xor(nHi, o);
        nLo += __shfl_xor(nLo, o);
    }
    if (lane == 0) {
        wtot[wave][0] = nHi;
        wtot[wave][1] = nLo;
    }
    __syncthreads();
    const int totHi = wtot[0][0] + wtot[1][0] + wtot[2][0] + wtot[3][0];
    const bool useLo = totHi <= 3;
    const int sel = useLo ? 1 : 0;
    const int t = useLo ? tLo : tHi;
    int off = 0;
    for (int w2 = 0; w2 < wave; w2++) off += wtot[w2][sel];
    int total = wtot[0][sel] + wtot[1][sel] + wtot[2][sel] + wtot[3][sel];
    if (total > cd.cand_cap) {
        if (tid == 0) atomicOr(err, 1);
        total = cd.cand_cap;
    }
    for (int r = r_begin; r < r_end; r++)
        for (int xb = 0; xb < cw; xb += 64) {
            const int x = xb + lane;
            bool fl = false;
            uint32_t m = 0;
            if (x < cw) {
                const uint8_t* p = scb + (r + 1) * spitch + x + 1;
                m = p[0];
                fl = (int)m > t && m > nb_max8(p, spitch);
            }
            const unsigned long long mask = __ballot(fl);
            if (fl) {
                const int pos = off + lane_prefix(mask);
                if (pos < cd.cand_cap) out[pos] = ((m - 1u) << 24) | ((uint32_t)(r + 3) << 12) | (uint32_t)(x + 3);
            }
            off += __popcll(mask);
        }
    if (tid == 0) *out_total = total;
}

// LDS bytes the kernel needs for cells of up to cw x ch examined pixels (worst-case tile alignment).
size_t fast_cell_lds_bytes(int cw, int ch, int ccap, int* qcap_out) {
    const int pitch = (15 + cw + 6 + 15) & ~15;
    const int spitch = (cw + 2 + 3) & ~3;
    const int qcap = FC_ROWS_PER_ITER * 256;  // 4 rows x 64 lanes x 4 pixels
    if (qcap_out) *qcap_out = qcap;
    return (size_t)pitch * (ch + 6) + (size_t)((spitch * (ch + 2) + 15) & ~15) + (size_t)4 * qcap * 2 + (size_t)4 * ccap * 4;
}

static int corner_list_cap() {
    static int cap = -1;
    if (cap < 0) {
        const char* e = getenv("ORBHIP_FC_CORNER_CAP");  // tests shrink it to reach the dense fallback
        cap = e ? atoi(e) : 512;
        if (cap < 1) cap = 1;
    }
    return cap;
}

// Tile columns and rows are stored in 8 bits each and one lane owns one 4-pixel group of a row.
bool fast_cell_supported(int cw, int ch) {
    return 15 + cw + 6 <= 256 && ch <= 249 && fast_cell_lds_bytes(cw, ch, corner_list_cap(), nullptr) <= 64 * 1024;
}

void launch_fast_cell(hipStream_t st, const GeomDev& hg, int batch, int max_cw, int max_ch, const uint8_t* pyr,
                      const GeomDev* dg, const CellDesc* cells, uint32_t* cand, int* n_total, int* err) {
    int qcap = 0;
    const int ccap = corner_list_cap();
    const size_t lds = fast_cell_lds_bytes(max_cw, max_ch, ccap, &qcap);
    dim3 grid(hg.cells_per_frame, batch), block(256);
    hipLaunchKernelGGL(k_fast_cell, grid, block, lds, st, pyr, dg, cells, cand, n_total, err, qcap, ccap);
}

}  // namespace orbk
