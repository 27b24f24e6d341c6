// Fused per-cell FAST for gfx950: one workgroup per (cell, frame) does what the reference does per
// cell in src/ORBextractor.cc:599-614 — cv::FAST(cellImage, kps, fastTh, true), and again with
// threshold 7 when that found <= 3 keypoints — entirely out of LDS:
//
//   1. the cell image (examined pixels + 3-px halo) is staged into LDS with aligned 16-byte loads;
//   2. QUICK TEST (every wave on its own rows): every examined pixel is tested on the four compass
//      pixels of the Bresenham circle (a 9-arc always contains one of {0,8} and one of {4,12}), a
//      necessary condition for a corner at the threshold of the pass, two pixels per packed 16-bit
//      instruction; survivors are appended to the wave's own list (ballot prefix sums);
//   3. MEASURE (whole 64-entry chunks of the wave's list, dense lanes): the exact corner measure M = max(A, B)
//      (fast_measure.h) of each listed pixel goes to an LDS score tile that is zero elsewhere;
//      pixels with M above the threshold are appended to a corner list;
//   4. NMS over the corner list: a pixel is a keypoint at threshold t iff M > t and M is a strict
//      maximum of its 8 neighbours inside the cell (non-corners and out-of-cell pixels count as 0,
//      which a strict maximum of the raw measure reproduces because M > t >= any non-corner's
//      measure).  Keypoints set a bit in a bitmap (one word = 32 pixels of one row);
//   5. ORDERED OUTPUT: popcounts of the bitmap words, a block-wide exclusive scan and a bit walk
//      emit the keypoints in raster order, whatever order the lists were filled in.
//      If the corner list overflows (very dense texture) step 4 scans the score tile densely.
// Steps 2-5 run with threshold fastTh; if that leaves <= 3 keypoints the score tile and bitmap are
// cleared and the steps run again with threshold 7, as the reference calls FAST a second time.
//
// Pyramid bytes are read from HBM once; no score plane is written.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "extract_kernels.h"
#include "fast_measure.h"

namespace orbk {

#define FC_THREADS 512
#define FC_WAVES (FC_THREADS / 64)
#define FC_WCAP (2 * 256 + 64)  // per-wave list: survivors of two rows + a carried-over partial chunk
#define FC_QCAP (FC_WAVES * FC_WCAP)

// Developer instrumentation (-DORBHIP_FC_PHASE_TIMING, never in a shipped build): shader-clock cycles every
// wave spends in each phase of k_fast_cell, summed over waves; read and reset through orbhip_debug_fc_phase.
#ifdef ORBHIP_FC_PHASE_TIMING
__device__ unsigned long long fc_phase[8];
#define FC_T0 unsigned long long tprev_ = __builtin_readcyclecounter();
#define FC_T(k)                                                              \
    {                                                                        \
        const unsigned long long now_ = __builtin_readcyclecounter();        \
        if (lane == 0) atomicAdd(&fc_phase[k], now_ - tprev_);               \
        tprev_ = now_;                                                       \
    }
#else
#define FC_T0
#define FC_T(k)
#endif

__device__ __forceinline__ int lane_prefix(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// Inclusive prefix sum over the 64 lanes with DPP row shifts and broadcasts (no LDS).
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2, 3
    return v;
}

// maximum of the 8 neighbours of the score byte at p (score tile has a zero border)
__device__ __forceinline__ uint32_t nb_max8(const uint8_t* __restrict__ p, int sp) {
    // two-operand 16-bit maxima issue at twice the rate of the 32-bit ones
    const uint32_t a = max_u16(max_u16(p[-sp - 1], p[-sp]), p[-sp + 1]);
    const uint32_t b = max_u16(max_u16(p[sp - 1], p[sp]), p[sp + 1]);
    return max_u16(max_u16(a, b), max_u16(p[-1], p[1]));
}

// Compass quick test of the four pixels of group gg in examined row r (a pixel can only be a corner
// if, of the circle pixels straight above / below and left / right, one of each pair is brighter than
// v + t, or one of each pair darker than v - t), with 32-bit add / subtract / logic only (the
// instructions that issue at full rate on this chip), two pixels per operation: the even pixels (0, 2) and the odd pixels (1, 3) of a
// group are kept as two 16-bit fields of one register.  With BT = 0x8000 + v + t per field, bit 15 of
// BT - I says "I is not brighter than v + t"; with DK = 0x8000 + v - t - 1, bit 15 of DK - I says "I is
// darker than v - t".  No field ever borrows from its neighbour (every difference stays within
// 0x8000 +- 511).  A pixel passes iff (up or down) and (left or right) are brighter, or the same with
// darker.  row points at the group's dword in its tile row.  Returns the four pass bits of the group (bit j = pixel j).
__device__ __forceinline__ uint32_t quick_row_swar(const uint32_t* __restrict__ row, int pitch, uint32_t T0, uint32_t T1) {
    const uint32_t wl = row[-1], wm = row[0], wr = row[1];
    const uint32_t up = *(const uint32_t*)((const uint8_t*)row - 3 * pitch);
    const uint32_t dn = *(const uint32_t*)((const uint8_t*)row + 3 * pitch);
    const uint32_t M = 0x00ff00ffu, H = 0x80008000u;
    const uint32_t ce = wm & M, co = (wm >> 8) & M;
    const uint32_t ue = up & M, uo = (up >> 8) & M;
    const uint32_t de = dn & M, dd = (dn >> 8) & M;
    const uint32_t le = (wl >> 8) & M;                                        // left of pixels 0, 2: bytes 1, 3 of wl
    const uint32_t lo = __builtin_amdgcn_alignbit(ce, wl & M, 16);            // left of pixels 1, 3: wl byte 2, wm byte 0
    const uint32_t re = __builtin_amdgcn_alignbit((wr >> 8) & M, co, 16);     // right of pixels 0, 2: wm byte 3, wr byte 1
    const uint32_t ro = wr & M;                                               // right of pixels 1, 3: wr bytes 0, 2
#define QSWAR(C, U, D, L, R, P)                                                              \
    {                                                                                        \
        const uint32_t ch = (C) | H;                                                         \
        const uint32_t bt = ch + T0, dk = ch - T1;                                           \
        const uint32_t notbright = ((bt - (U)) & (bt - (D))) | ((bt - (L)) & (bt - (R)));    \
        const uint32_t dark = ((dk - (U)) | (dk - (D))) & ((dk - (L)) | (dk - (R)));         \
        P = (notbright ^ H) | dark;                                                          \
    }
    uint32_t pe, po;
    QSWAR(ce, ue, de, le, re, pe);
    QSWAR(co, uo, dd, lo, ro, po);
#undef QSWAR
    // bit 15 / 31 of pe: pixels 0 / 2; of po: pixels 1 / 3
    const uint32_t x = ((pe >> 15) & 0x00010001u) | ((po >> 14) & 0x00020002u);
    return (x | (x >> 14)) & 0xfu;
}

// Everything about a cell that does not depend on the frame, worked out once per image size on the host
// (build_fast_cell_recs) so that a workgroup starts its tile loads after one scalar load of this record
// instead of a few hundred scalar and vector instructions of index arithmetic per wave.
struct FastCellRec {
    long long src_off;    // byte offset, inside a frame's pyramid, of the tile's first (16-byte aligned) byte
    long long cand_off;   // offset (keys) of the cell's candidate list inside a frame's arena
    int cw, ch;           // examined pixels (cw <= 0 or ch <= 0: nothing to do)
    int stride;           // row stride of the level's padded plane
    int th, pitch, pc;    // tile rows, tile row pitch in bytes (multiple of 16), pitch / 16
    int spitch;           // row pitch of the score tile
    int bw, nwords;       // bitmap words per row, bitmap words
    uint32_t bw_inv;      // ceil(65536 / bw)
    int c_lo, g_first;    // first examined tile column, its 4-pixel group
    int ngroups;          // 4-pixel groups per examined row
    uint32_t ng_inv;      // floor(2^32 / ngroups) + 1: i / ngroups == mulhi(i, ng_inv) for i < 2^16 (ngroups >= 2)
    int iters;            // quick-test iterations per wave
    int dr, dg;           // quotient and remainder of 64 * FC_WAVES / ngroups
    int step_addr, wrap_addr;
    uint32_t lowbits, highbits;  // pixels of a row's first / last group outside the examined columns
    int scb_off, bm_off, qlist_off, clist_off;  // LDS byte offsets
    int nz;               // 16-byte units to clear per pass (score tile + bitmap)
    int cand_cap;
    int per;              // bitmap words per thread in the output phase
    uint32_t cw_inv;      // floor(2^32 / cw) + 1 for cw >= 2
};

// WPS = waves per SIMD the register allocation aims at: the workgroups a CU holds are limited by their LDS
// (cell size), and the register budget follows from that number (8 waves per SIMD = 64 VGPRs for the small
// cells of which four workgroups fit, 6 = 80 VGPRs for the large ones of which three fit).
template <int WPS>
__global__ __launch_bounds__(FC_THREADS) __attribute__((amdgpu_waves_per_eu(WPS, WPS))) void k_fast_cell(
    const uint8_t* __restrict__ pyr, const FastCellRec* __restrict__ recs, long long frame_bytes, long long cand_per_frame,
    int cells_per_frame, int fast_th, uint32_t* __restrict__ cand, int* __restrict__ n_total, int* __restrict__ err, int ccap) {
    extern __shared__ uint4 lds4[];
    __shared__ int cn;
    __shared__ int red[FC_WAVES];
    int c = blockIdx.x, f = blockIdx.y;
    xcd_frame_swizzle(c, f, (int)gridDim.x, (int)gridDim.y);
    const FastCellRec cd = recs[c];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    int* out_total = n_total + (long long)f * cells_per_frame + c;
    if (cd.cw <= 0 || cd.ch <= 0) {
        if (tid == 0) *out_total = 0;
        return;
    }
    FC_T0
    const int stride = cd.stride;
    const int cw = cd.cw, ch = cd.ch;
    const int th = cd.th, pitch = cd.pitch;
    const int spitch = cd.spitch;
    const int bw = cd.bw, nwords = cd.nwords;
    const uint32_t bw_inv = cd.bw_inv;
    uint8_t* imgb = (uint8_t*)lds4;
    uint8_t* scb = imgb + cd.scb_off;  // (ch + 2) rows of spitch bytes, zero border
    uint32_t* bm = (uint32_t*)(imgb + cd.bm_off);
    uint16_t* qlist = (uint16_t*)(imgb + cd.qlist_off);
    uint16_t* clist = (uint16_t*)(imgb + cd.clist_off);

    {
        // 16 lanes per tile row (pitch <= 256 bytes = 16 chunks), 32 rows per sweep of the block and
        // up to four rows in flight per lane; no index division
        const int pc = cd.pc;
        const int cj = tid & 15, r0 = tid >> 4;
        if (cj < pc) {
            const uint8_t* src = pyr + (long long)f * frame_bytes + cd.src_off + 16 * cj;
            uint4* dst = (uint4*)imgb + cj;
            for (int r = r0; r < th; r += 128) {
                // rows past the end are clamped to the last row (same bytes stored twice), so that
                // the four loads are unconditional and issue back to back
                uint4 v[4];
                int rr[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    rr[k] = min(r + 32 * k, th - 1);
                    v[k] = *(const uint4*)(src + (long long)rr[k] * stride);
                }
#pragma unroll
                for (int k = 0; k < 4; k++) dst[rr[k] * pc] = v[k];
            }
        }
    }

    FC_T(0)
    const int tHi = min(max(fast_th, 0), 255), tLo = 7;
    const int c_lo = cd.c_lo;  // examined tile columns [c_lo, c_lo + cw)
    const int g_first = cd.g_first;
    // A row of the cell is ngroups 4-pixel groups wide; the groups of the cell, row after row, form one sequence
    // that is cut into chunks of 64, one group per lane, so that every lane has work whatever the cell's width.
    // A wave takes chunk (8 it + wave) of the first half of the sequence and the corresponding chunk of the
    // second half in iteration it (two chunks per prefix sum); a lane's (row, group) advances by 512 groups per
    // iteration with the quotient and remainder of 512 / ngroups, which are wave-uniform.
    const int ngroups = cd.ngroups;
    const int iters = cd.iters;
    const int dr = cd.dr, dg = cd.dg;
    const int step_addr = cd.step_addr, wrap_addr = cd.wrap_addr;
    const uint32_t lowbits = cd.lowbits, highbits = cd.highbits;
    uint32_t* out = cand + (long long)f * cand_per_frame + cd.cand_off;

    // One FAST call of the reference on the cell image at threshold t.  Written as a body that is expanded twice
    // (not as a loop over the two thresholds): a loop makes everything the later phases need loop-invariant, the
    // compiler computes all of it up front and spills dozens of scalar registers to hold it.
    auto fast_pass = [&](const int t, const bool first) -> bool {
    {
        uint4* z128 = (uint4*)scb;  // score tile and bitmap are contiguous
        const int nz = cd.nz;
        for (int i = tid; i < nz; i += FC_THREADS) z128[i] = make_uint4(0, 0, 0, 0);
        if (tid == 0) cn = 0;
    }
    __syncthreads();
    FC_T(1)
    const uint32_t T0 = (uint32_t)t * 0x00010001u, T1 = T0 + 0x00010001u;

    // ---- quick test + exact measure: every wave works through its own rows and its own list, no
    // block barrier until all rows are done.  A wave tests two rows, appends the survivors to its
    // list and then measures whole chunks of 64 entries taken from the end of the list, so the
    // measure always runs with all lanes busy; what is left (< 64 entries) waits for the next rows.
    uint16_t* wl = qlist + wave * FC_WCAP;
    int n = 0;
    // per-lane position in the two halves of the group sequence
    int rowA, glA, rowB, glB;
    {
        const int iA = (wave << 6) + lane, iB = ((FC_WAVES * iters + wave) << 6) + lane;
        // 2^32 / 1 does not fit the reciprocal: one group per row means row = index
        rowA = ngroups == 1 ? iA : (int)__umulhi((uint32_t)iA, cd.ng_inv);
        glA = iA - rowA * ngroups;
        rowB = ngroups == 1 ? iB : (int)__umulhi((uint32_t)iB, cd.ng_inv);
        glB = iB - rowB * ngroups;
    }
    int adrA = (rowA + 3) * pitch + 4 * (g_first + glA), adrB = (rowB + 3) * pitch + 4 * (g_first + glB);
#define FC_VNIB(gl) (0xfu ^ (lowbits & (uint32_t)(((gl)-1) >> 31)) ^ (highbits & (uint32_t)((ngroups - 2 - (gl)) >> 31)))
#define FC_ADVANCE(row, gl, adr)                                   \
    {                                                              \
        const int t_ = (gl) + dg - ngroups, m_ = t_ >> 31;         \
        gl = t_ + (ngroups & m_);                                  \
        row += dr + 1 + m_;                                        \
        adr += step_addr + (wrap_addr & ~m_);                      \
    }
    for (int it = 0; it < iters; it++) {
        const bool haveB = __builtin_amdgcn_readfirstlane(rowB) < ch;  // lane 0 holds the chunk's first group
        // per-lane pass nibbles (bit j = pixel j of the group passes and is an examined pixel)
        uint32_t nibA = 0, nibB = 0;
        if (rowA < ch) nibA = quick_row_swar((const uint32_t*)(imgb + adrA), pitch, T0, T1) & FC_VNIB(glA);
        if (haveB && rowB < ch) nibB = quick_row_swar((const uint32_t*)(imgb + adrB), pitch, T0, T1) & FC_VNIB(glB);
        // every lane appends its own survivors as one run per chunk: positions from one wave-wide prefix sum
        // of the per-lane counts (both chunks packed in one register).  The order of a wave's list does not
        // matter: the bitmap restores raster order.
        const uint32_t cnt = (uint32_t)__popc(nibA) | ((uint32_t)__popc(nibB) << 16);
        const uint32_t incl = wave_inclusive_sum(cnt);
        const uint32_t excl = incl - cnt;
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const int totA = (int)(tot & 0xffffu), totB = (int)(tot >> 16);
        {
            int pos = n + (int)(excl & 0xffffu);
            const uint32_t e = (uint32_t)(rowA << 8) | (uint32_t)(4 * (g_first + glA));
#pragma unroll
            for (int j = 0; j < 4; j++)
                if ((nibA >> j) & 1u) wl[pos++] = (uint16_t)(e + j);
        }
        if (haveB) {
            int pos = n + totA + (int)(excl >> 16);
            const uint32_t e = (uint32_t)(rowB << 8) | (uint32_t)(4 * (g_first + glB));
#pragma unroll
            for (int j = 0; j < 4; j++)
                if ((nibB >> j) & 1u) wl[pos++] = (uint16_t)(e + j);
        }
        FC_ADVANCE(rowA, glA, adrA);
        FC_ADVANCE(rowB, glB, adrB);
        n += totA + totB;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        FC_T(2)
        const bool last = it + 1 == iters;
        while (n >= 64 || (last && n > 0)) {
            const int i = n - 64 + lane;  // the last (partial) chunk leaves the low lanes idle
            uint32_t e = 0;
            int m = 0;
            if (i >= 0) {
                e = wl[i];
                m = min(fast_measure_tile(imgb + ((e >> 8) + 3) * pitch + (e & 255), pitch), 255);
            }
            const bool corner = m > t;
            if (corner) scb[((e >> 8) + 1) * spitch + ((e & 255) - c_lo + 1)] = (uint8_t)m;
            const unsigned long long mask = __ballot(corner);
            int cbase = 0;
            if (lane == 0 && mask) cbase = atomicAdd(&cn, __popcll(mask));
            cbase = __builtin_amdgcn_readfirstlane(cbase);
            if (corner) {
                const int pos = cbase + lane_prefix(mask);
                if (pos < ccap) clist[pos] = (uint16_t)e;
            }
            n = max(n - 64, 0);
        }
        FC_T(3)
    }
#undef FC_VNIB
#undef FC_ADVANCE
    __syncthreads();
    FC_T(4)

    // ---- strict-maximum NMS into the bitmap --------------------------------------------------------
    const int ncorner = cn;
    if (ncorner <= ccap) {
        for (int i = tid; i < ncorner; i += FC_THREADS) {
            const uint32_t e = clist[i];
            const int r = e >> 8, x = (int)(e & 255) - c_lo;
            const uint8_t* p = scb + (r + 1) * spitch + x + 1;
            if ((uint32_t)p[0] > nb_max8(p, spitch)) atomicOr(&bm[r * bw + (x >> 5)], 1u << (x & 31));
        }
    } else {
        // corner list overflowed: visit every examined pixel
        for (int p0 = tid; p0 < cw * ch; p0 += FC_THREADS) {
            const int r = cw == 1 ? p0 : (int)__umulhi((uint32_t)p0, cd.cw_inv), x = p0 - r * cw;
            const uint8_t* p = scb + (r + 1) * spitch + x + 1;
            const uint32_t m = p[0];
            if (m != 0 && m > nb_max8(p, spitch)) atomicOr(&bm[r * bw + (x >> 5)], 1u << (x & 31));
        }
    }
    __syncthreads();
    FC_T(5)

    // ---- count, then raster-ordered output --------------------------------------------------------
    const int per = cd.per;
    const int w0 = min(tid * per, nwords), w1 = min(w0 + per, nwords);
    int mine = 0;
    for (int w = w0; w < w1; w++) mine += __popc(bm[w]);
    const int incl = (int)wave_inclusive_sum((uint32_t)mine);
    if (lane == 63) red[wave] = incl;
    __syncthreads();
    int off = incl - mine, total = 0;
#pragma unroll
    for (int w = 0; w < FC_WAVES; w++) {
        if (w < wave) off += red[w];
        total += red[w];
    }
    FC_T(6)
    // "if(cellKeyPoints[i][j].size()<=3)" -> clear and FAST(cellImage, ..., 7, true)
    if (first && total <= 3) {
        __syncthreads();
        return false;
    }
    if (total > cd.cand_cap) {
        if (tid == 0) atomicOr(err, 1);
        total = cd.cand_cap;
    }
    for (int w = w0; w < w1; w++) {
        uint32_t bits = bm[w];
        // w / bw by multiplication: bw <= 8 and w < 2048, so the rounding error stays below 1 / bw
        const int r = (int)(((uint32_t)w * bw_inv) >> 16), xb = (w - r * bw) << 5;
        while (bits) {
            const int b = __ffs(bits) - 1;
            bits &= bits - 1;
            const int x = xb + b;
            if (off < cd.cand_cap) {
                const uint32_t sv = (uint32_t)scb[(r + 1) * spitch + x + 1] - 1u;
                // cell-image coordinates: x - iniX = x + 3, y - iniY = r + 3
                out[off] = (sv << 24) | ((uint32_t)(r + 3) << 12) | (uint32_t)(x + 3);
            }
            off++;
        }
    }
    if (tid == 0) *out_total = total;
    FC_T(7)
    return true;
    };
    if (!fast_pass(tHi, true)) fast_pass(tLo, false);
}

// LDS layout of one cell: tile | score tile | bitmap | per-wave survivor lists | corner list.
struct FastCellLayout {
    int pitch, th, spitch, bw, nwords, scb_off, bm_off, qlist_off, clist_off, total;
};
static FastCellLayout fast_cell_layout(int shift, int cw, int ch, int ccap) {
    FastCellLayout l;
    l.pitch = (shift + cw + 6 + 15) & ~15;
    l.th = ch + 6;
    l.spitch = (cw + 2 + 3) & ~3;
    l.bw = (cw + 31) >> 5;
    l.nwords = l.bw * ch;
    l.scb_off = l.pitch * l.th;
    l.bm_off = l.scb_off + ((l.spitch * (ch + 2) + 15) & ~15);
    l.qlist_off = (l.bm_off + l.nwords * 4 + 15) & ~15;  // cleared with 16-byte stores together with the score tile
    l.clist_off = l.qlist_off + FC_QCAP * 2;
    l.total = l.clist_off + ccap * 2;
    return l;
}

// LDS bytes the kernel needs for cells of up to cw x ch examined pixels (worst-case tile alignment).
size_t fast_cell_lds_bytes(int cw, int ch, int ccap) { return (size_t)fast_cell_layout(15, cw, ch, ccap).total; }

static int corner_list_cap() {
    static int cap = -1;
    if (cap < 0) {
        const char* e = getenv("ORBHIP_FC_CORNER_CAP");  // tests shrink it to reach the dense NMS path
        cap = e ? atoi(e) : 2048;
        if (cap < 1) cap = 1;
    }
    return cap;
}

// Tile columns and rows are stored in 8 bits each and one lane owns one 4-pixel group of a row.
bool fast_cell_supported(int cw, int ch) {
    return 15 + cw + 6 <= 256 && ch <= 249 && fast_cell_lds_bytes(cw, ch, corner_list_cap()) <= 64 * 1024;
}

size_t fast_cell_rec_bytes() { return sizeof(FastCellRec); }

// One record per cell of the per-frame cell table, in table order.
void build_fast_cell_recs(const GeomDev& hg, const CellDesc* cells, int ncells, void* out) {
    FastCellRec* recs = (FastCellRec*)out;
    const int ccap = corner_list_cap();
    for (int i = 0; i < ncells; i++) {
        const CellDesc& cd = cells[i];
        FastCellRec r;
        memset(&r, 0, sizeof(r));
        r.cw = cd.cw;
        r.ch = cd.ch;
        if (cd.cw > 0 && cd.ch > 0) {
            const LevelGeom& L = hg.lv[cd.level];
            const int gx0 = ORB_EDGE + cd.x0 - 3;  // padded-plane column of the tile's first needed byte
            const int gxs = gx0 & ~15;
            const int shift = gx0 - gxs;
            const int gy0 = ORB_EDGE + cd.y0 - 3;
            const FastCellLayout l = fast_cell_layout(shift, cd.cw, cd.ch, ccap);
            r.src_off = L.plane_off + (long long)gy0 * L.stride + gxs;
            r.cand_off = cd.cand_off;
            r.stride = L.stride;
            r.th = l.th;
            r.pitch = l.pitch;
            r.pc = l.pitch >> 4;
            r.spitch = l.spitch;
            r.bw = l.bw;
            r.nwords = l.nwords;
            r.bw_inv = (65536u + (uint32_t)l.bw - 1u) / (uint32_t)l.bw;
            r.c_lo = shift + 3;
            r.g_first = r.c_lo >> 2;
            r.ngroups = ((r.c_lo + cd.cw + 3) >> 2) - r.g_first;
            r.ng_inv = r.ngroups >= 2 ? (uint32_t)(0x100000000ULL / (uint32_t)r.ngroups) + 1u : 0u;
            const int nchunks = (r.ngroups * cd.ch + 63) >> 6;
            r.iters = (nchunks + 2 * FC_WAVES - 1) / (2 * FC_WAVES);
            r.dr = (64 * FC_WAVES) / r.ngroups;
            r.dg = 64 * FC_WAVES - r.dr * r.ngroups;
            r.step_addr = r.dr * l.pitch + 4 * r.dg;
            r.wrap_addr = l.pitch - 4 * r.ngroups;
            r.lowbits = (1u << (r.c_lo & 3)) - 1u;
            r.highbits = 0xfu & ~((1u << (r.c_lo + cd.cw - 4 * (r.g_first + r.ngroups - 1))) - 1u);
            r.scb_off = l.scb_off;
            r.bm_off = l.bm_off;
            r.qlist_off = l.qlist_off;
            r.clist_off = l.clist_off;
            r.nz = (l.qlist_off - l.scb_off) / 16;
            r.cand_cap = cd.cand_cap;
            r.per = (l.nwords + FC_THREADS - 1) / FC_THREADS;
            r.cw_inv = cd.cw >= 2 ? (uint32_t)(0x100000000ULL / (uint32_t)cd.cw) + 1u : 0u;
        }
        recs[i] = r;
    }
}

void launch_fast_cell(hipStream_t st, const GeomDev& hg, int batch, int max_cw, int max_ch, const uint8_t* pyr,
                      const void* recs, uint32_t* cand, int* n_total, int* err) {
    const int ccap = corner_list_cap();
    const size_t lds = fast_cell_lds_bytes(max_cw, max_ch, ccap);
    dim3 grid(hg.cells_per_frame, batch), block(FC_THREADS);
    const FastCellRec* r = (const FastCellRec*)recs;
    // four workgroups per CU need <= 40 KB each and 8 waves per SIMD
    if (lds <= 40 * 1024)
        hipLaunchKernelGGL(k_fast_cell<8>, grid, block, lds, st, pyr, r, hg.frame_bytes, hg.cand_per_frame, hg.cells_per_frame,
                           hg.fast_th, cand, n_total, err, ccap);
    else
        hipLaunchKernelGGL(k_fast_cell<6>, grid, block, lds, st, pyr, r, hg.frame_bytes, hg.cand_per_frame, hg.cells_per_frame,
                           hg.fast_th, cand, n_total, err, ccap);
}

}  // namespace orbk

#ifdef ORBHIP_FC_PHASE_TIMING
extern "C" int orbhip_debug_fc_phase(unsigned long long* out8) {
    unsigned long long z[8] = {0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(orbk::fc_phase), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(orbk::fc_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
