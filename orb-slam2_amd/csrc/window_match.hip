// Feature grid of a frame and the windowed matchers of the reference that draw their candidates
// from it, for gfx950, batched over independent frames / frame pairs:
//   Frame grid               src/Frame.cc:78-79,110-124,268-278 (include/Frame.h:37-38,90-92)
//   Frame::GetFeaturesInArea src/Frame.cc:201-266
//   ORBmatcher::WindowSearch src/ORBmatcher.cc:420-527
//   ORBmatcher::SearchForInitialization  src/ORBmatcher.cc:609-724
//
// Both matchers are sequential in the reference: whether keypoint i1 of F1 may take feature i2 of
// F2 depends on what earlier keypoints took.  The work splits into
//   1. candidate generation + Hamming distances, independent per keypoint of F1 (one lane each):
//      count pass, per-frame exclusive scan, fill pass -> CSR lists of (i2, distance) in the exact
//      order GetFeaturesInArea returns them;
//   2. resolution, one wave per frame pair, walking the keypoints of F1 in order with the
//      reference's state (vMatchedDistance / vnMatches21 / vnMatches12) in LDS; the candidates of
//      one keypoint are examined by the 64 lanes at once and reduced with the ordered-minimum rule
//      (first candidate in list order attaining the minimum; second order statistic with
//      multiplicity); the candidate lists of runs of consecutive keypoints are staged in LDS first;
//   3. rotation histogram, ComputeThreeMaxima (src/ORBmatcher.cc:1759-1800) and the final filter.
// All float expressions are evaluated operation by operation in fp32 (no contraction), as the
// reference's are.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <vector>

#include "matcher_internal.h"

namespace {

#define GRID_COLS ORBHIP_FRAME_GRID_COLS
#define GRID_ROWS ORBHIP_FRAME_GRID_ROWS
#define GRID_CELLS ORBHIP_FRAME_GRID_CELLS
#define HISTO_LENGTH 30
#define TH_HIGH 100
#define TH_LOW 20
#define NONE16 0xffffu
#define KEY_NONE 0xffffffffffffffffull
#define RB_CAP 4096  // candidate entries staged in LDS per run of the resolution kernel

struct GridGeom {
    float min_x, min_y;  // (float)mnMinX, (float)mnMinY
    float inv_w, inv_h;  // mfGridElementWidthInv, mfGridElementHeightInv
};

struct KpDev {
    float x, y, size, angle, response;
    int32_t octave, class_id;
};

// std::round(float): half away from zero
__device__ __forceinline__ float round_half_away(float v) {
    const float t = truncf(v);
    const float fr = fabsf(__fsub_rn(v, t));
    return fr >= 0.5f ? (v < 0.f ? t - 1.f : t + 1.f) : t;
}

// Frame::PosInGrid (src/Frame.cc:268-278); -1 if the keypoint is in no cell
__device__ __forceinline__ int pos_in_grid(const GridGeom& gg, float x, float y) {
    const int posX = (int)round_half_away(__fmul_rn(__fsub_rn(x, gg.min_x), gg.inv_w));
    const int posY = (int)round_half_away(__fmul_rn(__fsub_rn(y, gg.min_y), gg.inv_h));
    if (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) return -1;
    return posX * GRID_ROWS + posY;
}

// ---------------------------------------------------------------------------------------------
// Grid build: one workgroup per frame.  Keys (cell << 13 | keypoint index) are sorted in LDS
// (bitonic network; keys are distinct, so the order is the reference's push_back order inside
// every cell), then the cell starts are lower bounds in the sorted keys.
#define GB_THREADS 256
__global__ __launch_bounds__(GB_THREADS) void k_frame_grid(const KpDev* __restrict__ kp, const int32_t* __restrict__ n_arr,
                                                           int cap, GridGeom gg, int32_t* __restrict__ cell_start,
                                                           int32_t* __restrict__ cell_items) {
    extern __shared__ uint32_t keys[];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n = min(max(n_arr[f], 0), cap);
    int np2 = 64;
    while (np2 < n) np2 <<= 1;
    const KpDev* fk = kp + (long long)f * cap;
    for (int i = tid; i < np2; i += GB_THREADS) {
        uint32_t key = 0xffffffffu;
        if (i < n) {
            const int c = pos_in_grid(gg, fk[i].x, fk[i].y);
            if (c >= 0) key = ((uint32_t)c << 13) | (uint32_t)i;
        }
        keys[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (np2 >> 1); t += GB_THREADS) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // index with bit j clear
                const int hi = lo | j;
                const uint32_t a = keys[lo], b = keys[hi];
                const bool up = (lo & k) == 0;
                if ((a > b) == up) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
            __syncthreads();
        }
    int32_t* cs = cell_start + (long long)f * (GRID_CELLS + 1);
    for (int c = tid; c <= GRID_CELLS; c += GB_THREADS) {
        const uint32_t bound = (uint32_t)c << 13;  // first key of cell c; cell GRID_CELLS bounds the valid keys
        int lo = 0, hi = np2;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (keys[mid] < bound) lo = mid + 1;
            else hi = mid;
        }
        cs[c] = lo;
    }
    int32_t* it = cell_items + (long long)f * cap;
    for (int i = tid; i < n; i += GB_THREADS) {
        const uint32_t key = keys[i];
        if (key != 0xffffffffu) it[i] = (int32_t)(key & 0x1fffu);
    }
}

// ---------------------------------------------------------------------------------------------
// Candidate enumeration of one GetFeaturesInArea call.  Cells (ix, iyMin..iyMax) of one grid
// column are consecutive in the CSR arrays, so a column is one contiguous run of items.
struct AreaQuery {
    float x, y, r;
    int min_level, max_level;
    bool active;
};

// Sixteen lanes enumerate one query together: every lane of the group takes one item of the
// column's run per step, the group's slice of the wave ballot gives the count and the ordered
// position of each hit.  FILL = false only counts; FILL = true calls emit(i2, position) for every
// hit, positions counting from `o` in the reference's order.  All lanes of the wave must call this
// together (inactive queries included).  Returns the number of hits of the lane's group.
#define AREA_LPQ 16
template <bool FILL, class Emit>
__device__ __forceinline__ int area_group(const GridGeom& gg, const KpDev* __restrict__ kp, const int32_t* __restrict__ cs,
                                          const int32_t* __restrict__ items, const AreaQuery& q, int li, int o, Emit emit) {
    int nMinCellX = 0, nMaxCellX = -1, nMinCellY = 0, nMaxCellY = -1;
    if (q.active) {
        const float fx = __fsub_rn(q.x, gg.min_x), fy = __fsub_rn(q.y, gg.min_y);
        nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(fx, q.r), gg.inv_w)));
        nMaxCellX = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(fx, q.r), gg.inv_w)));
        nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(fy, q.r), gg.inv_h)));
        nMaxCellY = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(fy, q.r), gg.inv_h)));
        // "if(nMinCellX>=FRAME_GRID_COLS) return" ... "if(nMaxCellY<0) return": empty ranges below
        if (nMinCellX >= GRID_COLS || nMaxCellX < 0 || nMinCellY >= GRID_ROWS || nMaxCellY < 0) nMaxCellX = nMinCellX - 1;
        if (nMinCellY > nMaxCellY) nMaxCellX = nMinCellX - 1;
    }
    const bool bCheckLevels = !(q.min_level == -1 && q.max_level == -1);
    const bool bSameLevel = bCheckLevels && q.min_level == q.max_level;
    const int shift = (int)(threadIdx.x & 63u & ~(AREA_LPQ - 1));  // first lane of the group inside the wave
    int hits = 0;
    for (int ix = nMinCellX; __ballot(ix <= nMaxCellX); ix++) {
        int j0 = 0, j1 = 0;
        if (ix <= nMaxCellX) {
            j0 = cs[ix * GRID_ROWS + nMinCellY];
            j1 = cs[ix * GRID_ROWS + nMaxCellY + 1];
        }
        for (int base = j0; __ballot(base < j1); base += AREA_LPQ) {
            const int j = base + li;
            bool hit = false;
            int i2 = 0;
            if (j < j1) {
                i2 = items[j];
                const KpDev& k = kp[i2];
                hit = true;
                if (bCheckLevels && !bSameLevel) {
                    if (k.octave < q.min_level || k.octave > q.max_level) hit = false;
                } else if (bSameLevel) {
                    if (k.octave != q.min_level) hit = false;
                }
                if (fabsf(__fsub_rn(k.x, q.x)) > q.r || fabsf(__fsub_rn(k.y, q.y)) > q.r) hit = false;
            }
            const uint32_t slice = (uint32_t)(__ballot(hit) >> shift) & ((1u << AREA_LPQ) - 1u);
            if (FILL && hit) emit(i2, o + hits + __popc(slice & ((1u << li) - 1u)));
            hits += __popc(slice);
        }
    }
    return hits;
}

// How the queries of a frame are defined.
enum { MODE_EXPLICIT = 0, MODE_SEARCH_INIT = 1, MODE_WINDOW = 2, MODE_PROBE = 3 };
struct QuerySource {
    int mode;
    const OrbAreaQuery* queries;  // MODE_EXPLICIT: nq per frame
    const OrbProbe* probes;       // MODE_PROBE: nq per frame, n1[f] of them in use
    const KpDev* kp1;             // the matchers: keypoints of F1 (cap per frame)
    const int32_t* n1;
    const float* prev;            // MODE_SEARCH_INIT: vbPrevMatched (cap * 2 floats per frame)
    const uint8_t* valid1;        // MODE_WINDOW
    float r;
    int min_level, max_level;     // MODE_WINDOW: minScaleLevel, maxScaleLevel
    int nq;                       // queries per frame (= cap for the matchers)
};

__device__ __forceinline__ AreaQuery make_query(const QuerySource& s, int f, int i) {
    AreaQuery q;
    q.active = false;
    q.x = q.y = q.r = 0.f;
    q.min_level = q.max_level = -1;
    if (s.mode == MODE_EXPLICIT) {
        const OrbAreaQuery& e = s.queries[(long long)f * s.nq + i];
        q.x = e.x;
        q.y = e.y;
        q.r = e.r;
        q.min_level = e.min_level;
        q.max_level = e.max_level;
        q.active = true;
        return q;
    }
    if (i >= s.n1[f]) return q;
    if (s.mode == MODE_PROBE) {
        const OrbProbe& e = s.probes[(long long)f * s.nq + i];
        // a projection that is not a number reaches GetFeaturesInArea in the reference, where the
        // float -> int conversions are undefined: such a probe is dropped here
        if (!e.valid || !(fabsf(e.u) <= 3.0e38f) || !(fabsf(e.v) <= 3.0e38f) || !(fabsf(e.r) <= 3.0e38f)) return q;
        q.x = e.u;
        q.y = e.v;
        q.r = e.r;
        q.min_level = e.min_level;
        q.max_level = e.max_level;
        q.active = true;
        return q;
    }
    const KpDev& k = s.kp1[(long long)f * s.nq + i];
    const int level1 = k.octave;
    q.r = s.r;
    q.min_level = q.max_level = level1;
    if (s.mode == MODE_SEARCH_INIT) {
        if (level1 > 0) return q;  // src/ORBmatcher.cc:626-627
        q.x = s.prev[((long long)f * s.nq + i) * 2];
        q.y = s.prev[((long long)f * s.nq + i) * 2 + 1];
        q.active = true;
    } else {
        if (!s.valid1[(long long)f * s.nq + i]) return q;          // :438-441
        if (s.min_level > 0 && level1 < s.min_level) return q;      // :431,446-448
        if (s.max_level < INT_MAX && level1 > s.max_level) return q;  // :432,450-452
        q.x = k.x;
        q.y = k.y;
        q.active = true;
    }
    return q;
}

__global__ __launch_bounds__(256) void k_area_count(QuerySource s, const KpDev* __restrict__ kp2, const int32_t* __restrict__ cell_start,
                                                    const int32_t* __restrict__ cell_items, int cap2, GridGeom gg,
                                                    int32_t* __restrict__ counts) {
    const int i = (blockIdx.x * 256 + threadIdx.x) / AREA_LPQ, li = threadIdx.x % AREA_LPQ, f = blockIdx.y;
    AreaQuery q;
    q.active = false;
    q.x = q.y = q.r = 0.f;
    q.min_level = q.max_level = -1;
    if (i < s.nq) q = make_query(s, f, i);
    const int c = area_group<false>(gg, kp2 + (long long)f * cap2, cell_start + (long long)f * (GRID_CELLS + 1),
                                    cell_items + (long long)f * cap2, q, li, 0, [](int, int) {});
    if (i < s.nq && li == 0) counts[(long long)f * (s.nq + 1) + i] = c;
}

// In-place exclusive scan of the nq counts of every frame (one workgroup per frame); entry nq
// receives the frame's total, which is also written to totals[f].
__global__ __launch_bounds__(256) void k_scan_counts(int32_t* __restrict__ counts, int nq, int32_t* __restrict__ totals) {
    __shared__ int red[4];
    __shared__ int carry;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t* c = counts + (long long)f * (nq + 1);
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nq; base += 256) {
        const int i = base + tid;
        const int v = i < nq ? c[i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) red[wave] = incl;
        __syncthreads();
        int off = carry;
        for (int w = 0; w < wave; w++) off += red[w];
        if (i < nq) c[i] = off + incl - v;
        __syncthreads();
        if (tid == 255) carry = off + incl;
        __syncthreads();
    }
    if (tid == 0) {
        c[nq] = carry;
        if (totals) totals[f] = carry;
    }
}

// Per-frame bases of the candidate lists inside the pair buffer, on the device (the asynchronous form of the
// matchers: nothing is read back by the host): exclusive scan of the frames' totals.  If the lists need more
// than `capacity` entries the call cannot run: every list is emptied (starts zeroed, bases -1, so that neither
// the fill pass nor the sequential stage touches the pair buffer), flag 8 is raised and the count needed is left
// for orbhip_matcher_synchronize to report.
__global__ __launch_bounds__(256) void k_frame_bases(const int32_t* __restrict__ totals, int batch, long long capacity,
                                                     long long* __restrict__ base, int32_t* __restrict__ starts, int nq,
                                                     int* __restrict__ err) {
    __shared__ long long carry;
    __shared__ long long red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < batch; b0 += 256) {
        const int f = b0 + tid;
        const long long v = f < batch ? totals[f] : 0;
        long long incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const long long t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) red[wave] = incl;
        __syncthreads();
        long long off = carry;
        for (int w = 0; w < wave; w++) off += red[w];
        if (f < batch) base[f] = off + incl - v;
        __syncthreads();
        if (tid == 255) carry = off + incl;
        __syncthreads();
    }
    const long long total = carry;
    if (total > capacity) {
        for (int f = tid; f < batch; f += 256) base[f] = -1;
        const long long n = (long long)batch * (nq + 1);
        for (long long i = tid; i < n; i += 256) starts[i] = 0;
        if (tid == 0) {
            atomicOr(err, 8);
            *(long long*)(err + 2) = total;
        }
    }
}

// Fill pass.  out_idx (explicit queries): indices only, frame f at f*out_cap, overflow flagged.
// out_pairs (matchers): (distance << 16 | i2) at frame_base[f] + start.
__global__ __launch_bounds__(256) void k_area_fill(QuerySource s, const KpDev* __restrict__ kp2, const int32_t* __restrict__ cell_start,
                                                   const int32_t* __restrict__ cell_items, int cap2, GridGeom gg,
                                                   const int32_t* __restrict__ starts, int32_t* __restrict__ out_idx,
                                                   int out_cap, int* __restrict__ err, const uint8_t* __restrict__ desc1,
                                                   const uint8_t* __restrict__ desc2, const long long* __restrict__ frame_base,
                                                   uint32_t* __restrict__ out_pairs) {
    const int i = (blockIdx.x * 256 + threadIdx.x) / AREA_LPQ, li = threadIdx.x % AREA_LPQ, f = blockIdx.y;
    AreaQuery q;
    q.active = false;
    q.x = q.y = q.r = 0.f;
    q.min_level = q.max_level = -1;
    if (i < s.nq) q = make_query(s, f, i);
    const int32_t* st = starts + (long long)f * (s.nq + 1);
    const int o = i < s.nq ? st[i] : 0;
    const KpDev* kf = kp2 + (long long)f * cap2;
    const int32_t* cs = cell_start + (long long)f * (GRID_CELLS + 1);
    const int32_t* it = cell_items + (long long)f * cap2;
    if (out_idx) {
        if (i == 0 && li == 0 && st[s.nq] > out_cap) atomicOr(err, 4);
        int32_t* dst = out_idx + (long long)f * out_cap;
        area_group<true>(gg, kf, cs, it, q, li, o, [&](int i2, int pos) {
            if (pos < out_cap) dst[pos] = i2;
        });
    } else {
        if (frame_base[f] < 0) return;  // the lists of this call did not fit the reservation
        uint32_t* dst = out_pairs + frame_base[f];
        uint4 d1a = make_uint4(0, 0, 0, 0), d1b = d1a;
        if (q.active) {
            const uint4* d1 = (const uint4*)(desc1 + ((long long)f * s.nq + i) * 32);
            d1a = d1[0];
            d1b = d1[1];
        }
        const uint8_t* d2f = desc2 + (long long)f * cap2 * 32;
        area_group<true>(gg, kf, cs, it, q, li, o, [&](int i2, int pos) {
            const uint4* t = (const uint4*)(d2f + (long long)i2 * 32);
            const uint4 a = t[0], b = t[1];
            const int dist = __popc(a.x ^ d1a.x) + __popc(a.y ^ d1a.y) + __popc(a.z ^ d1a.z) + __popc(a.w ^ d1a.w) +
                             __popc(b.x ^ d1b.x) + __popc(b.y ^ d1b.y) + __popc(b.z ^ d1b.z) + __popc(b.w ^ d1b.w);
            dst[pos] = ((uint32_t)dist << 16) | (uint32_t)i2;
        });
    }
}

// "Match to the most similar keypoint in the radius" (src/ORBmatcher.cc:1105-1124, :1234-1253, :1384-1404,
// :1469-1489): per probe the first feature, in GetFeaturesInArea order, that attains the least distance.  The
// probes of these methods (Fuse, SearchBySim3) do not interact — no feature is taken away by an earlier probe —
// so sixteen lanes enumerate one probe's area, keep their own least (distance, position, index) key and the
// group reduces it; no candidate lists are stored.
__global__ __launch_bounds__(256) void k_best_in_radius(QuerySource s, const KpDev* __restrict__ kp2,
                                                        const int32_t* __restrict__ cell_start,
                                                        const int32_t* __restrict__ cell_items, int cap2, GridGeom gg,
                                                        const uint8_t* __restrict__ desc1, const uint8_t* __restrict__ desc2,
                                                        int32_t* __restrict__ best_idx, int32_t* __restrict__ best_dist) {
    const int i = (blockIdx.x * 256 + threadIdx.x) / AREA_LPQ, li = threadIdx.x % AREA_LPQ, f = blockIdx.y;
    AreaQuery q;
    q.active = false;
    q.x = q.y = q.r = 0.f;
    q.min_level = q.max_level = -1;
    if (i < s.nq) q = make_query(s, f, i);
    uint4 d1a = make_uint4(0, 0, 0, 0), d1b = d1a;
    if (q.active) {
        const uint4* d1 = (const uint4*)(desc1 + ((long long)f * s.nq + i) * 32);
        d1a = d1[0];
        d1b = d1[1];
    }
    const uint8_t* d2f = desc2 + (long long)f * cap2 * 32;
    unsigned long long key = ~0ull;  // distance | position in the list | feature index
    area_group<true>(gg, kp2 + (long long)f * cap2, cell_start + (long long)f * (GRID_CELLS + 1), cell_items + (long long)f * cap2, q,
                     li, 0, [&](int i2, int pos) {
                         const uint4* t = (const uint4*)(d2f + (long long)i2 * 32);
                         const uint4 a = t[0], b = t[1];
                         const int dist = __popc(a.x ^ d1a.x) + __popc(a.y ^ d1a.y) + __popc(a.z ^ d1a.z) + __popc(a.w ^ d1a.w) +
                                          __popc(b.x ^ d1b.x) + __popc(b.y ^ d1b.y) + __popc(b.z ^ d1b.z) + __popc(b.w ^ d1b.w);
                         const unsigned long long k = ((unsigned long long)dist << 32) | ((unsigned long long)pos << 16) | (unsigned)i2;
                         key = k < key ? k : key;
                     });
#pragma unroll
    for (int o = AREA_LPQ / 2; o >= 1; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o);
        key = other < key ? other : key;
    }
    if (i < s.nq && li == 0) {
        const long long at = (long long)f * s.nq + i;
        best_idx[at] = key == ~0ull ? -1 : (int32_t)(key & 0xffffu);
        best_dist[at] = key == ~0ull ? INT_MAX : (int32_t)(key >> 32);
    }
}

// ---------------------------------------------------------------------------------------------
// Resolution: one wave per frame pair.
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// The two smallest of the wave's 32-bit keys (distinct or KEY32_NONE), by an inclusive scan with DPP
// row shifts and broadcasts whose combining step keeps the two smallest of two disjoint sets; lane
// 63 ends up with the result, which is broadcast.
#define KEY32_NONE 0xffffffffu
__device__ __forceinline__ void two_smallest_step(uint32_t& k1, uint32_t& k2, uint32_t p1, uint32_t p2) {
    const uint32_t lo = min(k1, p1), hi = max(k1, p1);
    k2 = min(hi, min(k2, p2));
    k1 = lo;
}
#define TS_STEP(ctrl, rowmask)                                                                            \
    {                                                                                                     \
        const uint32_t p1 = (uint32_t)__builtin_amdgcn_update_dpp((int)KEY32_NONE, (int)k1, ctrl, rowmask, 0xf, false); \
        const uint32_t p2 = (uint32_t)__builtin_amdgcn_update_dpp((int)KEY32_NONE, (int)k2, ctrl, rowmask, 0xf, false); \
        two_smallest_step(k1, k2, p1, p2);                                                                \
    }
__device__ __forceinline__ void wave_two_smallest32(uint32_t k1, uint32_t k2, uint32_t& o1, uint32_t& o2) {
    TS_STEP(0x111, 0xf)  // row_shr:1
    TS_STEP(0x112, 0xf)  // row_shr:2
    TS_STEP(0x114, 0xf)  // row_shr:4
    TS_STEP(0x118, 0xf)  // row_shr:8
    TS_STEP(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
    TS_STEP(0x143, 0xc)  // row_bcast:31 into rows 2 and 3
    o1 = (uint32_t)__builtin_amdgcn_readlane((int)k1, 63);
    o2 = (uint32_t)__builtin_amdgcn_readlane((int)k2, 63);
}
#undef TS_STEP

// bin of a match (src/ORBmatcher.cc:493-498, :679-684)
__device__ __forceinline__ int rot_bin(float a1, float a2) {
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = __fsub_rn(a1, a2);
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)round_half_away(__fmul_rn(rot, factor));
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

// ComputeThreeMaxima (src/ORBmatcher.cc:1759-1800) on the bin sizes; bit i of the result is set
// iff bin i is one of the (up to) three kept bins.
__device__ inline uint32_t three_maxima_mask(const int* hist) {
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < HISTO_LENGTH; i++) {
        const int sz = hist[i];
        if (sz > max1) {
            max3 = max2;
            max2 = max1;
            max1 = sz;
            ind3 = ind2;
            ind2 = ind1;
            ind1 = i;
        } else if (sz > max2) {
            max3 = max2;
            max2 = sz;
            ind3 = ind2;
            ind2 = i;
        } else if (sz > max3) {
            max3 = sz;
            ind3 = i;
        }
    }
    if ((float)max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
    } else if ((float)max3 < 0.1f * (float)max1) {
        ind3 = -1;
    }
    uint32_t mask = 0;
    if (ind1 >= 0) mask |= 1u << ind1;
    if (ind2 >= 0) mask |= 1u << ind2;
    if (ind3 >= 0) mask |= 1u << ind3;
    return mask;
}

// Sequential part of SearchForInitialization (src/ORBmatcher.cc:609-724): one wave per frame pair walks F1's
// keypoints over the candidate lists.  LDS: four uint16 arrays of cap entries, the candidate run buffer
// (RB_CAP entries, at byte offset 8 * cap) and the histogram.
__global__ __launch_bounds__(64) void k_resolve_init(const KpDev* __restrict__ kp1, const int32_t* __restrict__ n1_arr,
                                                     const KpDev* __restrict__ kp2, const int32_t* __restrict__ n2_arr, int cap,
                                                const int32_t* __restrict__ starts, const long long* __restrict__ frame_base,
                                                const uint32_t* __restrict__ pairs, float ratio, int check_ori,
                                                float* __restrict__ prev, int32_t* __restrict__ matches_out,
                                                int32_t* __restrict__ nmatches_out) {
    extern __shared__ uint16_t lds16[];
    __shared__ int hist[HISTO_LENGTH + 2];
    __shared__ int nm_sh;
    const int f = blockIdx.x, lane = threadIdx.x;
    const int n1 = min(max(n1_arr[f], 0), cap), n2 = min(max(n2_arr[f], 0), cap);
    uint16_t* matchedDist = lds16;      // vMatchedDistance; NONE16 = INT_MAX
    uint16_t* m21 = lds16 + cap;        // vnMatches21
    uint16_t* m12 = lds16 + 2 * cap;    // vnMatches12
    uint16_t* partner = lds16 + 3 * cap;  // F2 index i1 was pushed to the histogram with
    for (int i = lane; i < cap; i += 64) {
        matchedDist[i] = NONE16;
        m21[i] = NONE16;
        m12[i] = NONE16;
        partner[i] = NONE16;
    }
    if (lane < HISTO_LENGTH + 2) hist[lane] = 0;
    wave_fence();
    const int32_t* st = starts + (long long)f * (cap + 1);
    const uint32_t* fp = pairs + frame_base[f];
    int nmatches = 0;

    // Candidate lists are consumed in runs: the lists of as many consecutive keypoints as fit are copied
    // to LDS with one coalesced pass, so that the sequential walk below never waits on global memory.
    uint32_t* cbuf = (uint32_t*)(lds16 + 4 * cap);
    for (int c0 = 0; c0 < n1; c0 += 64) {
        // starts of 64 keypoints at once; lane j holds those of keypoint c0 + j
        const int ci = min(c0 + lane, n1);
        const int my_s = st[ci], my_e = st[min(ci + 1, cap)];
        const int chunk = min(64, n1 - c0);
        int j0 = 0;
        while (j0 < chunk) {
            const int S = __shfl(my_s, j0);
            // keypoints j0 .. j0 + q - 1 fit the buffer together (list ends are monotone)
            const unsigned long long fit = __ballot(lane >= j0 && lane < chunk && my_e - S <= RB_CAP);
            const int q = __popcll(fit);
            const bool buffered = q > 0;
            const int jn = buffered ? j0 + q : j0 + 1;
            if (buffered) {
                const int E = __shfl(my_e, jn - 1);
                for (int p = S + lane; p < E; p += 64) cbuf[p - S] = fp[p];
                wave_fence();
            }
            for (int j = j0; j < jn; j++) {
                const int i1 = c0 + j;
                const int s = __shfl(my_s, j), e = __shfl(my_e, j);
                if (s >= e) continue;
                // key: distance above, position in the list below (lists hold at most 8192 entries)
                uint32_t k1 = KEY32_NONE, k2 = KEY32_NONE;
                for (int b = s; b < e; b += 64) {
                    const int p = b + lane;
                    if (p < e) {
                        const uint32_t pr = buffered ? cbuf[p - S] : fp[p];
                        const uint32_t i2 = pr & 0xffffu, dist = pr >> 16;
                        if ((uint32_t)matchedDist[i2] > dist) {  // :648
                            const uint32_t key = (dist << 13) | (uint32_t)(p - s);
                            k2 = min(max(k1, key), k2);
                            k1 = min(k1, key);
                        }
                    }
                }
                uint32_t b1, b2;
                wave_two_smallest32(k1, k2, b1, b2);
                if (b1 == KEY32_NONE) continue;
                const int bestDist = (int)(b1 >> 13);
                const int bpos = s + (int)(b1 & 0x1fffu);
                const int bestIdx2 = (int)((buffered ? cbuf[bpos - S] : fp[bpos]) & 0xffffu);
                const float fBest2 = b2 == KEY32_NONE ? (float)INT_MAX : (float)(int)(b2 >> 13);
                if (bestDist <= TH_LOW && (float)bestDist < __fmul_rn(fBest2, ratio)) {  // :663-665
                    const uint32_t old = m21[bestIdx2];
                    if (old != NONE16) nmatches--;
                    wave_fence();
                    if (lane == 0) {
                        if (old != NONE16) m12[old] = NONE16;
                        m12[i1] = (uint16_t)bestIdx2;
                        m21[bestIdx2] = (uint16_t)i1;
                        matchedDist[bestIdx2] = (uint16_t)bestDist;
                        partner[i1] = (uint16_t)bestIdx2;
                    }
                    nmatches++;
                    wave_fence();
                }
            }
            wave_fence();  // the buffer is reused by the next run
            j0 = jn;
        }
    }
    wave_fence();

    // ---- rotation histogram, three maxima, filter -------------------------------------------------
    const KpDev* k1f = kp1 + (long long)f * cap;
    const KpDev* k2f = kp2 + (long long)f * cap;
    if (lane == 0) nm_sh = 0;
    wave_fence();
    if (check_ori) {
        for (int i1 = lane; i1 < n1; i1 += 64)
            if (partner[i1] != NONE16) atomicAdd(&hist[rot_bin(k1f[i1].angle, k2f[partner[i1]].angle)], 1);
        wave_fence();
        const uint32_t keep = three_maxima_mask(hist);
        int removed = 0;
        for (int i1 = lane; i1 < n1; i1 += 64)
            if (partner[i1] != NONE16 && !((keep >> rot_bin(k1f[i1].angle, k2f[partner[i1]].angle)) & 1u) &&
                m12[i1] != NONE16) {
                m12[i1] = NONE16;
                removed++;
            }
        if (removed) atomicAdd(&nm_sh, removed);
        wave_fence();
    }
    nmatches -= nm_sh;
    if (lane == 0) nmatches_out[f] = nmatches;
    int32_t* mo = matches_out + (long long)f * cap;
    for (int i1 = lane; i1 < cap; i1 += 64) {
        const uint32_t v = i1 < n1 ? m12[i1] : NONE16;
        mo[i1] = v == NONE16 ? -1 : (int32_t)v;
        if (v != NONE16) {  // "Update prev matched", :719-721
            prev[((long long)f * cap + i1) * 2] = k2f[v].x;
            prev[((long long)f * cap + i1) * 2 + 1] = k2f[v].y;
        }
    }
}

GridGeom make_geom(const OrbFrameBounds& b) {
    GridGeom g;
    g.min_x = (float)b.min_x;
    g.min_y = (float)b.min_y;
    g.inv_w = static_cast<float>(GRID_COLS) / static_cast<float>(b.max_x - b.min_x);  // src/Frame.cc:78
    g.inv_h = static_cast<float>(GRID_ROWS) / static_cast<float>(b.max_y - b.min_y);  // src/Frame.cc:79
    return g;
}

int check_frames(orbhip_matcher* m, int batch, int cap, const OrbFrameBounds& b) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (batch < 0 || cap < 1) return orb_fail(ORBHIP_EINVAL, "bad batch / capacity");
    if (cap > ORBHIP_FRAME_MAX_KEYPOINTS)
        return orb_fail(ORBHIP_ECAP, "cap %d > %d keypoints per frame", cap, ORBHIP_FRAME_MAX_KEYPOINTS);
    if (b.max_x <= b.min_x || b.max_y <= b.min_y) return orb_fail(ORBHIP_EINVAL, "empty image bounds");
    return ORBHIP_OK;
}

int ensure_work(orbhip_matcher* m, size_t bytes) {
    if (bytes <= m->work_bytes) return ORBHIP_OK;
    if (hipStreamSynchronize(m->stream) != hipSuccess) return orb_fail(ORBHIP_EHIP, "stream synchronisation failed");
    if (m->d_work) hipFree(m->d_work);
    m->d_work = nullptr;
    m->work_bytes = 0;
    if (hipMalloc(&m->d_work, bytes) != hipSuccess) return orb_fail(ORBHIP_ENOMEM, "hipMalloc of %zu bytes failed", bytes);
    m->work_bytes = bytes;
    return ORBHIP_OK;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Candidate stage shared by all matchers: count pass, per-frame scan, fill pass.  On return the
// workspace holds starts [batch*(nq+1)], the per-frame bases and the (distance << 16 | i2) lists.
struct Candidates {
    int32_t* starts;
    long long* d_base;
    uint32_t* pairs;
};

int build_candidates(orbhip_matcher* m, const QuerySource& s, const uint8_t* d_desc1, const KpDev* d_kp2, const uint8_t* d_desc2,
                     const int32_t* d_cell_start2, const int32_t* d_cell_items2, int batch, int cap2, const GridGeom& gg,
                     Candidates& out) {
    int rc;
    const int nq = s.nq;
    // workspace: starts [batch*(nq+1)] | totals [batch] | frame_base [batch] (int64) | pairs
    const size_t off_tot = align256(sizeof(int32_t) * (size_t)batch * (nq + 1));
    const size_t off_base = off_tot + align256(sizeof(int32_t) * (size_t)batch);
    const size_t off_pairs = off_base + align256(sizeof(long long) * (size_t)batch);
    if ((rc = ensure_work(m, off_pairs + sizeof(uint32_t) * (size_t)m->cand_reserved))) return rc;
    dim3 qgrid((nq * AREA_LPQ + 255) / 256, batch);
    {
        int32_t* starts = (int32_t*)m->d_work;
        int32_t* totals = (int32_t*)((char*)m->d_work + off_tot);
        hipLaunchKernelGGL(k_area_count, qgrid, dim3(256), 0, m->stream, s, d_kp2, d_cell_start2, d_cell_items2, cap2, gg, starts);
        hipLaunchKernelGGL(k_scan_counts, dim3(batch), dim3(256), 0, m->stream, starts, nq, totals);
    }
    if (m->cand_reserved > 0) {
        // asynchronous form: the reservation is the capacity, bases come from a device scan, nothing is read back
        const long long capacity = m->cand_reserved;  // the workspace holds at least that many past the headers
        out.starts = (int32_t*)m->d_work;
        out.d_base = (long long*)((char*)m->d_work + off_base);
        out.pairs = (uint32_t*)((char*)m->d_work + off_pairs);
        hipLaunchKernelGGL(k_frame_bases, dim3(1), dim3(256), 0, m->stream, (const int32_t*)((char*)m->d_work + off_tot), batch,
                           capacity, out.d_base, out.starts, nq, m->d_err);
        hipLaunchKernelGGL(k_area_fill, qgrid, dim3(256), 0, m->stream, s, d_kp2, d_cell_start2, d_cell_items2, cap2, gg,
                           out.starts, (int32_t*)nullptr, 0, m->d_err, d_desc1, d_desc2, out.d_base, out.pairs);
        return ORBHIP_OK;
    }
    std::vector<int32_t> totals(batch);
    HIPCHK(hipMemcpyAsync(totals.data(), (char*)m->d_work + off_tot, sizeof(int32_t) * batch, hipMemcpyDeviceToHost,
                          m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    std::vector<long long> base(batch);
    long long total = 0;
    for (int f = 0; f < batch; f++) {
        base[f] = total;
        total += totals[f];
    }
    // growing the workspace moves it: keep what the count pass produced
    const size_t need = off_pairs + sizeof(uint32_t) * (size_t)std::max<long long>(total, 1);
    if (need > m->work_bytes) {
        void* old = m->d_work;
        const size_t old_bytes = m->work_bytes;
        m->d_work = nullptr;
        m->work_bytes = 0;
        if ((rc = ensure_work(m, need + need / 4))) {
            hipFree(old);
            return rc;
        }
        hipError_t ce = hipMemcpy(m->d_work, old, std::min(old_bytes, off_pairs), hipMemcpyDeviceToDevice);
        hipFree(old);
        if (ce != hipSuccess) return orb_fail(ORBHIP_EHIP, "workspace copy failed: %s", hipGetErrorString(ce));
    }
    out.starts = (int32_t*)m->d_work;
    out.d_base = (long long*)((char*)m->d_work + off_base);
    out.pairs = (uint32_t*)((char*)m->d_work + off_pairs);
    HIPCHK(hipMemcpyAsync(out.d_base, base.data(), sizeof(long long) * batch, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));  // `base` is a stack-lifetime host buffer
    hipLaunchKernelGGL(k_area_fill, qgrid, dim3(256), 0, m->stream, s, d_kp2, d_cell_start2, d_cell_items2, cap2, gg, out.starts,
                       (int32_t*)nullptr, 0, m->d_err, d_desc1, d_desc2, out.d_base, out.pairs);
    return ORBHIP_OK;
}

enum { RULE_LOCAL_MAP = 0, RULE_RATIO = 1, RULE_BEST = 2 };  // acceptance rules of the two-stage resolution below
int run_probes(orbhip_matcher* m, const QuerySource& s, int pcap, const uint8_t* d_desc1, const KpDev* d_kp2, const uint8_t* d_desc2,
               const int32_t* d_n2, const int32_t* d_cs, const int32_t* d_ci, int batch, int cap, const GridGeom& gg,
               const uint8_t* d_taken, int rule, float ratio, int th_dist, int check_ori, int32_t* d_out, int32_t* d_nmatches);

// Shared body of SearchForInitialization and WindowSearch.
int run_matcher(orbhip_matcher* m, int mode, const OrbKeypoint* d_kp1, const uint8_t* d_desc1, const int32_t* d_n1,
                const uint8_t* d_valid1, const OrbKeypoint* d_kp2, const uint8_t* d_desc2, const int32_t* d_n2,
                const int32_t* d_cell_start2, const int32_t* d_cell_items2, int batch, int cap, OrbFrameBounds bounds,
                float* d_prev, int window_size, int min_level, int max_level, float ratio, int check_ori,
                int32_t* d_matches, int32_t* d_nmatches) {
    int rc = check_frames(m, batch, cap, bounds);
    if (rc) return rc;
    if (batch == 0) return ORBHIP_OK;
    if (!d_kp1 || !d_desc1 || !d_n1 || !d_kp2 || !d_desc2 || !d_n2 || !d_cell_start2 || !d_cell_items2 || !d_matches ||
        !d_nmatches || (mode == MODE_SEARCH_INIT && !d_prev) || (mode == MODE_WINDOW && !d_valid1))
        return orb_fail(ORBHIP_EINVAL, "null buffer");
    HIPCHK(hipSetDevice(m->device));
    (void)hipGetLastError();
    const GridGeom gg = make_geom(bounds);
    QuerySource s;
    s.mode = mode;
    s.queries = nullptr;
    s.probes = nullptr;
    s.kp1 = (const KpDev*)d_kp1;
    s.n1 = d_n1;
    s.prev = d_prev;
    s.valid1 = d_valid1;
    s.r = (float)window_size;  // int windowSize binds to "const float &r"
    s.min_level = min_level;
    s.max_level = max_level;
    s.nq = cap;
    HIPCHK(hipEventRecord(m->ev0, m->stream));
    if (mode == MODE_WINDOW) {
        // WindowSearch (src/ORBmatcher.cc:436-523) has the shape of the projection matchers: F1's keypoints in order,
        // features of F2 that already hold a map point skipped, best / second best with the ratio rule of :487, the
        // rotation histogram over the matches — its queries are F1's keypoints instead of probe records
        if ((rc = run_probes(m, s, cap, d_desc1, (const KpDev*)d_kp2, d_desc2, d_n2, d_cell_start2, d_cell_items2, batch, cap, gg,
                             nullptr, RULE_RATIO, ratio, TH_HIGH, check_ori, d_matches, d_nmatches)))
            return rc;
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(m->ev1, m->stream));
        return ORBHIP_OK;
    }
    Candidates c;
    if ((rc = build_candidates(m, s, d_desc1, (const KpDev*)d_kp2, d_desc2, d_cell_start2, d_cell_items2, batch, cap, gg, c)))
        return rc;
    const size_t lds = sizeof(uint16_t) * 4 * (size_t)cap + sizeof(uint32_t) * RB_CAP;
    HIPCHK(hipFuncSetAttribute((const void*)k_resolve_init, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_resolve_init, dim3(batch), dim3(64), lds, m->stream, (const KpDev*)d_kp1, d_n1, (const KpDev*)d_kp2,
                       d_n2, cap, c.starts, c.d_base, c.pairs, ratio, check_ori, d_prev, d_matches, d_nmatches);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(m->ev1, m->stream));
    if (m->cand_reserved == 0) HIPCHK(hipStreamSynchronize(m->stream));
    return ORBHIP_OK;
}

// Resolution of the projection matchers, in two stages.  A feature of the frame that already holds a map point,
// on entry or from an earlier probe, is skipped (src/ORBmatcher.cc:94, :389, :578, :1573, :1698), which makes the
// walk over the probes sequential — but all an earlier probe can do to a later one is take one of its candidates
// away.  So stage one (k_probe_topk, all probes in parallel, 16 lanes each) enumerates the probe's area as
// GetFeaturesInArea does and keeps, per probe, its PK best candidates among the features free on entry, in the
// reference's order (distance, then position in the list), and the count of its free candidates; stage two
// (k_resolve_probe, one wave per frame) walks the probes in order and only looks at those few: the first of them
// still free is the best, the next one the second best.  Only when fewer free ones are left than the rule needs
// AND the probe has more candidates than the PK listed does the wave enumerate the probe's area again and scan it
// as the reference does (rare: it takes that many earlier probes having taken exactly this probe's best candidates).
// No candidate lists are stored.
//   RULE_LOCAL_MAP  :101-124   best and second with their levels; accepted if best <= TH_HIGH unless
//                               both are on one level and best > ratio * second
//   RULE_RATIO      :585-600   (float)best <= (float)second * ratio && best <= TH_HIGH
//   RULE_BEST       :1580-1590, :1705-1715, :401-413   best <= th_dist
#define TAKEN16 0xfffeu
#define PK 4

// ascending compare-exchange
__device__ __forceinline__ void cmpx64(unsigned long long& a, unsigned long long& b) {
    const unsigned long long lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo;
    b = hi;
}

__device__ __forceinline__ int hamming256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
           __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// Candidate keys of the projection matchers: distance, then position in the GetFeaturesInArea list (the order in
// which the reference's strict "<" keeps candidates), with the feature index riding along in the low bits.
#define PKEY_NONE 0xffffffffffffffffull
__device__ __forceinline__ unsigned long long pkey(int dist, int pos, int i2) {
    return ((unsigned long long)dist << 29) | ((unsigned long long)pos << 16) | (unsigned)i2;
}
__device__ __forceinline__ uint32_t pkey_word(unsigned long long k) {  // distance << 16 | feature
    return k == PKEY_NONE ? KEY32_NONE : (uint32_t)((k >> 29) << 16) | (uint32_t)(k & 0xffffu);
}

// topk[(f*pcap + i)*PK + k] = (distance << 16 | feature) of probe i's k-th best candidate among the features free on
// entry, KEY32_NONE past the end; nfree[f*pcap + i] = number of its candidates that are free on entry.  The probe's
// area is enumerated as GetFeaturesInArea walks it; nothing is stored but these five words.
__global__ __launch_bounds__(256) void k_probe_topk(QuerySource s, const KpDev* __restrict__ kp2,
                                                    const int32_t* __restrict__ cell_start,
                                                    const int32_t* __restrict__ cell_items, int cap2, GridGeom gg,
                                                    const uint8_t* __restrict__ desc1, const uint8_t* __restrict__ desc2,
                                                    const uint8_t* __restrict__ taken, uint32_t* __restrict__ topk,
                                                    int32_t* __restrict__ nfree) {
    const int i = (blockIdx.x * 256 + threadIdx.x) / AREA_LPQ, li = threadIdx.x % AREA_LPQ, f = blockIdx.y;
    AreaQuery q;
    q.active = false;
    q.x = q.y = q.r = 0.f;
    q.min_level = q.max_level = -1;
    if (i < s.nq) q = make_query(s, f, i);
    uint4 d1a = make_uint4(0, 0, 0, 0), d1b = d1a;
    if (q.active) {
        const uint4* d1 = (const uint4*)(desc1 + ((long long)f * s.nq + i) * 32);
        d1a = d1[0];
        d1b = d1[1];
    }
    const uint8_t* d2f = desc2 + (long long)f * cap2 * 32;
    const uint8_t* tk = taken ? taken + (long long)f * cap2 : nullptr;
    unsigned long long k0 = PKEY_NONE, k1 = PKEY_NONE, k2 = PKEY_NONE, k3 = PKEY_NONE;
    int cnt = 0;
    area_group<true>(gg, kp2 + (long long)f * cap2, cell_start + (long long)f * (GRID_CELLS + 1), cell_items + (long long)f * cap2, q,
                     li, 0, [&](int i2, int pos) {
                         if (tk && tk[i2]) return;
                         cnt++;
                         const uint4* t = (const uint4*)(d2f + (long long)i2 * 32);
                         unsigned long long x = pkey(hamming256(t[0], t[1], d1a, d1b), pos, i2);
                         cmpx64(k0, x);
                         cmpx64(k1, x);
                         cmpx64(k2, x);
                         k3 = x < k3 ? x : k3;
                     });
    // merge the 16 lanes' sorted quadruples: the four smallest of two sorted quadruples are min(a_i, b_{3-i}), a
    // bitonic sequence that two exchange stages sort
#pragma unroll
    for (int o = AREA_LPQ / 2; o >= 1; o >>= 1) {
        const unsigned long long b0 = __shfl_xor(k0, o), b1 = __shfl_xor(k1, o), b2 = __shfl_xor(k2, o), b3 = __shfl_xor(k3, o);
        k0 = b3 < k0 ? b3 : k0;
        k1 = b2 < k1 ? b2 : k1;
        k2 = b1 < k2 ? b1 : k2;
        k3 = b0 < k3 ? b0 : k3;
        cmpx64(k0, k2);
        cmpx64(k1, k3);
        cmpx64(k0, k1);
        cmpx64(k2, k3);
        cnt += __shfl_xor(cnt, o);
    }
    if (i < s.nq && li < PK) {
        const unsigned long long key = li == 0 ? k0 : li == 1 ? k1 : li == 2 ? k2 : k3;
        topk[((long long)f * s.nq + i) * PK + li] = pkey_word(key);
        if (li == 0) nfree[(long long)f * s.nq + i] = cnt;
    }
}

// Stage two takes the probes of a frame 64 at a time, one per lane, each as if it were the next one; a probe whose
// listed candidates include a feature that an earlier probe of the same pass takes (or that needs the full scan) ends
// the pass: the probes before it are settled, it is resolved by the whole wave, and the rest of the chunk goes again.
// LDS: owner of every feature (uint16: free, taken on entry, or probe index), the claims of a pass, the histogram.
template <int RULE>
__global__ __launch_bounds__(64) void k_resolve_probe(QuerySource src, int pcap, const KpDev* __restrict__ kp2,
                                                      const int32_t* __restrict__ n2_arr,
                                                      int cap2, const int32_t* __restrict__ cell_start,
                                                      const int32_t* __restrict__ cell_items, GridGeom gg,
                                                      const uint8_t* __restrict__ desc1, const uint8_t* __restrict__ desc2,
                                                      const uint32_t* __restrict__ topk, const int32_t* __restrict__ nfree,
                                                      const uint8_t* __restrict__ taken, float ratio, int th_dist, int check_ori,
                                                      int32_t* __restrict__ matches_out, int32_t* __restrict__ nmatches_out) {
    extern __shared__ uint16_t lds16[];
    __shared__ int hist[HISTO_LENGTH + 2];
    __shared__ int nm_sh;
    const int f = blockIdx.x, lane = threadIdx.x;
    const int np = min(max(src.n1[f], 0), pcap), n2 = min(max(n2_arr[f], 0), cap2);
    uint16_t* owner = lds16;
    uint32_t* claim = (uint32_t*)(lds16 + ((cap2 + 1) & ~1));  // per feature: lowest lane of the current pass that takes it
    for (int i = lane; i < cap2; i += 64) {
        owner[i] = (taken && i < n2 && taken[(long long)f * cap2 + i]) ? TAKEN16 : NONE16;
        claim[i] = KEY32_NONE;
    }
    if (lane < HISTO_LENGTH + 2) hist[lane] = 0;
    wave_fence();
    const int32_t* cs = cell_start + (long long)f * (GRID_CELLS + 1);
    const int32_t* its = cell_items + (long long)f * cap2;
    const uint8_t* d2f = desc2 + (long long)f * cap2 * 32;
    const KpDev* k2f = kp2 + (long long)f * cap2;
    // the angle a probe brings to the rotation histogram: the probe record's, or that of F1's keypoint (WindowSearch)
    auto probe_angle = [&](int ip) -> float {
        return src.mode == MODE_PROBE ? src.probes[(long long)f * pcap + ip].angle : src.kp1[(long long)f * pcap + ip].angle;
    };
    const uint4* tkf = (const uint4*)(topk + (long long)f * pcap * PK);
    const int32_t* nff = nfree + (long long)f * pcap;
    const bool need2 = RULE != RULE_BEST;
    // the acceptance rule of the overload on the best / second-best words (distance << 16 | feature)
    auto accepts = [&](uint32_t b1, uint32_t b2) -> bool {
        const int bestDist = (int)(b1 >> 16);
        if (RULE == RULE_LOCAL_MAP) {
            bool accept = bestDist <= TH_HIGH;
            if (accept && b2 != KEY32_NONE) {
                const int bestDist2 = (int)(b2 >> 16);
                if (k2f[b1 & 0xffffu].octave == k2f[b2 & 0xffffu].octave && (float)bestDist > __fmul_rn(ratio, (float)bestDist2))
                    accept = false;
            }
            return accept;
        } else if (RULE == RULE_RATIO) {
            const float fBest2 = b2 == KEY32_NONE ? (float)INT_MAX : (float)(int)(b2 >> 16);
            return (float)bestDist <= __fmul_rn(fBest2, ratio) && bestDist <= TH_HIGH;
        }
        return bestDist <= th_dist;
    };
    // Outcome of a probe from its listed candidates alone: 1 accept b1, 0 no match, -1 the whole list is needed.
    // nf: listed candidates still free (b1 the first, b2 the second); wlast: the last listed word.  With one free
    // listed candidate and more in the list, every unlisted candidate is at least as far as the last listed one, and
    // the rules are monotone in the second-best distance: if the best passes against that bound it passes against
    // the true second best, whatever its level.
    auto decide = [&](uint32_t b1, uint32_t b2, int nf, bool listed_all, uint32_t wlast) -> int {
        if (nf == 0) return listed_all ? 0 : -1;
        if (!need2 || nf >= 2 || listed_all) return accepts(b1, b2) ? 1 : 0;
        const int bestDist = (int)(b1 >> 16);
        if (bestDist > TH_HIGH) return 0;
        const float bound = (float)(int)(wlast >> 16);
        const bool passes = RULE == RULE_RATIO ? (float)bestDist <= __fmul_rn(bound, ratio) : !((float)bestDist > __fmul_rn(ratio, bound));
        return passes ? 1 : -1;
    };
    int nmatches = 0;
    for (int c0 = 0; c0 < np; c0 += 64) {
        const int ci = min(c0 + lane, np - 1);
        const uint4 my = tkf[ci];  // the probe's best free candidates, best first
        const uint32_t w[PK] = {my.x, my.y, my.z, my.w};
        const bool listed_all = nff[ci] <= PK;
        const int chunk = min(64, np - c0);
        int cur = 0;  // probes c0 .. c0 + cur - 1 of the chunk are settled
        while (cur < chunk) {
            // ---- every unsettled probe of the chunk at once, one per lane, as if it were the next one ----
            const bool live = lane >= cur && lane < chunk;
            uint32_t b1 = KEY32_NONE, b2 = KEY32_NONE;
            int nf = 0, last = PK - 1;  // last: the listed candidates after it cannot change this probe's outcome
            if (live) {
#pragma unroll
                for (int k = 0; k < PK; k++) {
                    if (w[k] != KEY32_NONE && owner[w[k] & 0xffffu] == NONE16) {
                        if (nf == 0) {
                            b1 = w[k];
                            if (!need2) last = k;
                        } else if (nf == 1) {
                            b2 = w[k];
                            last = k;
                        }
                        nf++;
                    }
                }
            }
            // too few of the listed candidates left to tell: this probe needs the full scan
            const int verdict = live ? decide(b1, b2, nf, listed_all, w[PK - 1]) : 0;
            bool alone = verdict < 0;
            const bool take = verdict > 0;
            // an earlier probe of the chunk that takes one of this probe's listed candidates changes its outcome:
            // every taker claims its feature with its lane number, the lowest lane wins the claim
            if (take) atomicMin(&claim[b1 & 0xffffu], (uint32_t)lane);
            wave_fence();
            if (live && !alone) {
#pragma unroll
                for (int k = 0; k < PK; k++)
                    if (k <= last && w[k] != KEY32_NONE && claim[w[k] & 0xffffu] < (uint32_t)lane) alone = true;
            }
            const unsigned long long amask = __ballot(alone);
            const int stop = amask ? __ffsll((long long)amask) - 1 : chunk;  // probes before it are settled by this pass
            wave_fence();
            if (take) claim[b1 & 0xffffu] = KEY32_NONE;
            const bool commit = take && lane < stop;
            if (commit) owner[b1 & 0xffffu] = (uint16_t)(c0 + lane);
            nmatches += __popcll(__ballot(commit));
            wave_fence();
            cur = stop;
            if (cur >= chunk) break;
            // ---- probe `cur` on its own, by the whole wave, as the reference does ----
            {
                const int j = cur, ip = c0 + j;
                uint32_t s1w = KEY32_NONE, s2w = KEY32_NONE;
                // its listed candidates first (lanes 0 .. PK-1 look one owner up each); the whole list only if too
                // few of them are left and there are more
                const uint32_t v0 = (uint32_t)__builtin_amdgcn_readlane((int)my.x, j), v1 = (uint32_t)__builtin_amdgcn_readlane((int)my.y, j);
                const uint32_t v2 = (uint32_t)__builtin_amdgcn_readlane((int)my.z, j), v3 = (uint32_t)__builtin_amdgcn_readlane((int)my.w, j);
                const bool all_listed = __builtin_amdgcn_readlane((int)listed_all, j) != 0;
                const uint32_t word = lane == 0 ? v0 : lane == 1 ? v1 : lane == 2 ? v2 : v3;
                const bool is_free = lane < PK && word != KEY32_NONE && owner[word & 0xffffu] == NONE16;
                const uint32_t fmask = (uint32_t)__ballot(is_free) & ((1u << PK) - 1u);
                const int nfj = __popc(fmask);
                int verdict = -1;
                if (nfj > 0) {
                    const int first = __ffs(fmask) - 1;
                    s1w = first == 0 ? v0 : first == 1 ? v1 : first == 2 ? v2 : v3;
                    const uint32_t rest = fmask & (fmask - 1u);
                    if (rest) {
                        const int second = __ffs(rest) - 1;
                        s2w = second == 1 ? v1 : second == 2 ? v2 : v3;
                    }
                }
                verdict = decide(s1w, s2w, nfj, all_listed, v3);
                if (verdict < 0) {
                    s1w = s2w = KEY32_NONE;
                    // the reference's scan: the probe's area once more (the first sixteen lanes enumerate it), every
                    // feature that is free now
                    AreaQuery q;
                    q.active = false;
                    q.x = q.y = q.r = 0.f;
                    q.min_level = q.max_level = -1;
                    uint4 d1a = make_uint4(0, 0, 0, 0), d1b = d1a;
                    if (lane < AREA_LPQ) {
                        q = make_query(src, f, ip);  // active: it has listed candidates
                        const uint4* d1 = (const uint4*)(desc1 + ((long long)f * pcap + ip) * 32);
                        d1a = d1[0];
                        d1b = d1[1];
                    }
                    unsigned long long k1 = PKEY_NONE, k2 = PKEY_NONE;
                    area_group<true>(gg, k2f, cs, its, q, lane % AREA_LPQ, 0, [&](int i2, int pos) {
                        if (owner[i2] != NONE16) return;
                        const uint4* t = (const uint4*)(d2f + (long long)i2 * 32);
                        const unsigned long long key = pkey(hamming256(t[0], t[1], d1a, d1b), pos, i2);
                        const unsigned long long hi = key < k1 ? k1 : key;
                        k2 = hi < k2 ? hi : k2;
                        k1 = key < k1 ? key : k1;
                    });
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) {
                        const unsigned long long p1 = __shfl_xor(k1, o), p2 = __shfl_xor(k2, o);
                        const unsigned long long lo = k1 < p1 ? k1 : p1, hi = k1 < p1 ? p1 : k1;
                        const unsigned long long m2 = k2 < p2 ? k2 : p2;
                        k2 = hi < m2 ? hi : m2;
                        k1 = lo;
                    }
                    s1w = pkey_word(k1);
                    s2w = pkey_word(k2);
                    verdict = s1w != KEY32_NONE && accepts(s1w, s2w) ? 1 : 0;
                }
                if (verdict > 0) {
                    if (lane == 0) owner[s1w & 0xffffu] = (uint16_t)ip;
                    nmatches++;
                }
                wave_fence();
                cur = j + 1;
            }
        }
    }
    wave_fence();
    if (lane == 0) nm_sh = 0;
    wave_fence();
    if (check_ori) {  // :1592-1602 and :1608-1628
        for (int i2 = lane; i2 < n2; i2 += 64)
            if (owner[i2] < TAKEN16) atomicAdd(&hist[rot_bin(probe_angle(owner[i2]), k2f[i2].angle)], 1);
        wave_fence();
        const uint32_t keep = three_maxima_mask(hist);
        int removed = 0;
        for (int i2 = lane; i2 < n2; i2 += 64)
            if (owner[i2] < TAKEN16 && !((keep >> rot_bin(probe_angle(owner[i2]), k2f[i2].angle)) & 1u)) {
                owner[i2] = NONE16;
                removed++;
            }
        if (removed) atomicAdd(&nm_sh, removed);
        wave_fence();
    }
    nmatches -= nm_sh;
    if (lane == 0) nmatches_out[f] = nmatches;
    int32_t* mo = matches_out + (long long)f * cap2;
    for (int i2 = lane; i2 < cap2; i2 += 64) {
        const uint32_t v = i2 < n2 ? owner[i2] : NONE16;
        mo[i2] = v < TAKEN16 ? (int32_t)v : -1;
    }
}

// The two stages for `batch` frames: queries from `s` (probe records, or F1's keypoints for WindowSearch), pcap of them
// per frame with their descriptors at d_desc1.
int run_probes(orbhip_matcher* m, const QuerySource& s, int pcap, const uint8_t* d_desc1, const KpDev* d_kp2, const uint8_t* d_desc2,
               const int32_t* d_n2, const int32_t* d_cs, const int32_t* d_ci, int batch, int cap, const GridGeom& gg,
               const uint8_t* d_taken, int rule, float ratio, int th_dist, int check_ori, int32_t* d_out, int32_t* d_nmatches) {
    // per probe: PK candidate words and the count of its free candidates
    const size_t tk_words = (size_t)batch * pcap * (PK + 1);
    if (sizeof(uint32_t) * tk_words > m->topk_bytes) {
        HIPCHK(hipStreamSynchronize(m->stream));
        if (m->d_topk) (void)hipFree(m->d_topk);
        m->d_topk = nullptr;
        m->topk_bytes = 0;
        if (hipMalloc(&m->d_topk, sizeof(uint32_t) * tk_words) != hipSuccess)
            return orb_fail(ORBHIP_ENOMEM, "hipMalloc of %zu bytes failed", sizeof(uint32_t) * tk_words);
        m->topk_bytes = sizeof(uint32_t) * tk_words;
    }
    uint32_t* d_topk = (uint32_t*)m->d_topk;
    int32_t* d_nfree = (int32_t*)(d_topk + (size_t)batch * pcap * PK);
    hipLaunchKernelGGL(k_probe_topk, dim3((unsigned)(((long long)pcap * AREA_LPQ + 255) / 256), batch), dim3(256), 0, m->stream,
                       s, (const KpDev*)d_kp2, d_cs, d_ci, cap, gg, d_desc1, d_desc2, d_taken, d_topk, d_nfree);
    const size_t lds = sizeof(uint16_t) * (size_t)((cap + 1) & ~1) + sizeof(uint32_t) * (size_t)cap;
#define LAUNCH_PROBE(R)                                                                                                   \
    hipLaunchKernelGGL(k_resolve_probe<R>, dim3(batch), dim3(64), lds, m->stream, s, pcap, (const KpDev*)d_kp2, d_n2,  \
                       cap, d_cs, d_ci, gg, d_desc1, d_desc2, d_topk, d_nfree, d_taken, ratio,     \
                       th_dist, check_ori, d_out, d_nmatches)
    if (rule == RULE_LOCAL_MAP) LAUNCH_PROBE(RULE_LOCAL_MAP);
    else if (rule == RULE_RATIO) LAUNCH_PROBE(RULE_RATIO);
    else LAUNCH_PROBE(RULE_BEST);
#undef LAUNCH_PROBE
    return ORBHIP_OK;
}

}  // namespace

extern "C" {

int orbhip_frame_grid_device(orbhip_matcher* m, const OrbKeypoint* d_kp, const int32_t* d_n, int batch, int cap,
                             OrbFrameBounds bounds, int32_t* d_cell_start, int32_t* d_cell_items) {
    int rc = check_frames(m, batch, cap, bounds);
    if (rc) return rc;
    if (batch == 0) return ORBHIP_OK;
    if (!d_kp || !d_n || !d_cell_start || !d_cell_items) return orb_fail(ORBHIP_EINVAL, "null buffer");
    HIPCHK(hipSetDevice(m->device));
    (void)hipGetLastError();
    int np2 = 64;
    while (np2 < cap) np2 <<= 1;
    hipLaunchKernelGGL(k_frame_grid, dim3(batch), dim3(GB_THREADS), sizeof(uint32_t) * (size_t)np2, m->stream, (const KpDev*)d_kp,
                       d_n, cap, make_geom(bounds), d_cell_start, d_cell_items);
    HIPCHK(hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_features_in_area_device(orbhip_matcher* m, const OrbKeypoint* d_kp, const int32_t* d_n, int batch, int cap,
                                   OrbFrameBounds bounds, const int32_t* d_cell_start, const int32_t* d_cell_items,
                                   const OrbAreaQuery* d_queries, int nq, int32_t* d_out_start, int32_t* d_out,
                                   int out_cap) {
    int rc = check_frames(m, batch, cap, bounds);
    if (rc) return rc;
    if (nq < 0 || out_cap < 0) return orb_fail(ORBHIP_EINVAL, "negative size");
    if (batch == 0) return ORBHIP_OK;
    if (!d_kp || !d_n || !d_cell_start || !d_cell_items || !d_out_start || (nq && !d_queries) || (out_cap && !d_out))
        return orb_fail(ORBHIP_EINVAL, "null buffer");
    HIPCHK(hipSetDevice(m->device));
    (void)hipGetLastError();
    QuerySource s;
    s.mode = MODE_EXPLICIT;
    s.queries = d_queries;
    s.probes = nullptr;
    s.kp1 = nullptr;
    s.n1 = nullptr;
    s.prev = nullptr;
    s.valid1 = nullptr;
    s.r = 0.f;
    s.min_level = s.max_level = -1;
    s.nq = nq;
    const GridGeom gg = make_geom(bounds);
    dim3 qgrid((std::max(nq, 1) * AREA_LPQ + 255) / 256, batch);
    if (nq)
        hipLaunchKernelGGL(k_area_count, qgrid, dim3(256), 0, m->stream, s, (const KpDev*)d_kp, d_cell_start, d_cell_items, cap,
                           gg, d_out_start);
    hipLaunchKernelGGL(k_scan_counts, dim3(batch), dim3(256), 0, m->stream, d_out_start, nq, (int32_t*)nullptr);
    if (nq)
        hipLaunchKernelGGL(k_area_fill, qgrid, dim3(256), 0, m->stream, s, (const KpDev*)d_kp, d_cell_start, d_cell_items, cap,
                           gg, d_out_start, d_out, out_cap, m->d_err, (const uint8_t*)nullptr, (const uint8_t*)nullptr,
                           (const long long*)nullptr, (uint32_t*)nullptr);
    HIPCHK(hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_matcher_reserve_candidates(orbhip_matcher* m, long long entries) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (entries < 0 || entries > (1LL << 34)) return orb_fail(ORBHIP_EINVAL, "bad reservation");
    HIPCHK(hipSetDevice(m->device));
    if (entries > 0) {
        // room for the lists plus the per-call headers of the largest batch (starts, totals, bases)
        const size_t header = align256(sizeof(int32_t) * (size_t)4096 * (ORBHIP_FRAME_MAX_KEYPOINTS + 1));
        int rc = ensure_work(m, std::min<size_t>(header, (size_t)64 << 20) + sizeof(uint32_t) * (size_t)entries);
        if (rc) return rc;
    }
    m->cand_reserved = entries;
    return ORBHIP_OK;
}

int orbhip_search_for_initialization_device(orbhip_matcher* m, const OrbKeypoint* d_kp1, const uint8_t* d_desc1,
                                            const int32_t* d_n1, const OrbKeypoint* d_kp2, const uint8_t* d_desc2,
                                            const int32_t* d_n2, const int32_t* d_cell_start2,
                                            const int32_t* d_cell_items2, int batch, int cap, OrbFrameBounds bounds,
                                            float* d_prev_matched, int window_size, float nnratio,
                                            int check_orientation, int32_t* d_matches12, int32_t* d_nmatches) {
    return run_matcher(m, MODE_SEARCH_INIT, d_kp1, d_desc1, d_n1, nullptr, d_kp2, d_desc2, d_n2, d_cell_start2, d_cell_items2,
                       batch, cap, bounds, d_prev_matched, window_size, -1, INT_MAX, nnratio, check_orientation, d_matches12,
                       d_nmatches);
}

int orbhip_window_search_device(orbhip_matcher* m, const OrbKeypoint* d_kp1, const uint8_t* d_desc1,
                                const int32_t* d_n1, const uint8_t* d_valid1, const OrbKeypoint* d_kp2,
                                const uint8_t* d_desc2, const int32_t* d_n2, const int32_t* d_cell_start2,
                                const int32_t* d_cell_items2, int batch, int cap, OrbFrameBounds bounds,
                                int window_size, int min_level, int max_level, float nnratio, int check_orientation,
                                int32_t* d_matches21, int32_t* d_nmatches) {
    return run_matcher(m, MODE_WINDOW, d_kp1, d_desc1, d_n1, d_valid1, d_kp2, d_desc2, d_n2, d_cell_start2, d_cell_items2,
                       batch, cap, bounds, nullptr, window_size, min_level, max_level, nnratio, check_orientation,
                       d_matches21, d_nmatches);
}

int orbhip_search_by_projection_device(orbhip_matcher* m, const OrbProbe* d_probes, const uint8_t* d_probe_desc,
                                       const int32_t* d_nprobes, int probe_cap, const OrbKeypoint* d_kp2,
                                       const uint8_t* d_desc2, const int32_t* d_n2, const int32_t* d_cell_start2,
                                       const int32_t* d_cell_items2, int batch, int cap, OrbFrameBounds bounds,
                                       const uint8_t* d_taken, int rule, float nnratio, int th_dist, int check_orientation,
                                       int32_t* d_match_of_feature, int32_t* d_nmatches) {
    int rc = check_frames(m, batch, cap, bounds);
    if (rc) return rc;
    if (probe_cap < 1 || probe_cap > ORBHIP_MAX_PROBES)
        return orb_fail(ORBHIP_ECAP, "probe_cap %d outside 1..%d", probe_cap, ORBHIP_MAX_PROBES);
    if (rule != ORBHIP_PROJ_LOCAL_MAP && rule != ORBHIP_PROJ_RATIO && rule != ORBHIP_PROJ_BEST)
        return orb_fail(ORBHIP_EINVAL, "unknown acceptance rule %d", rule);
    if (batch == 0) return ORBHIP_OK;
    if (!d_probes || !d_probe_desc || !d_nprobes || !d_kp2 || !d_desc2 || !d_n2 || !d_cell_start2 || !d_cell_items2 ||
        !d_match_of_feature || !d_nmatches)
        return orb_fail(ORBHIP_EINVAL, "null buffer");
    HIPCHK(hipSetDevice(m->device));
    (void)hipGetLastError();
    const GridGeom gg = make_geom(bounds);
    QuerySource s;
    s.mode = MODE_PROBE;
    s.queries = nullptr;
    s.probes = d_probes;
    s.kp1 = nullptr;
    s.n1 = d_nprobes;
    s.prev = nullptr;
    s.valid1 = nullptr;
    s.r = 0.f;
    s.min_level = s.max_level = -1;
    s.nq = probe_cap;
    HIPCHK(hipEventRecord(m->ev0, m->stream));
    if ((rc = run_probes(m, s, probe_cap, d_probe_desc, (const KpDev*)d_kp2, d_desc2, d_n2, d_cell_start2, d_cell_items2, batch, cap,
                         gg, d_taken, rule == ORBHIP_PROJ_LOCAL_MAP ? RULE_LOCAL_MAP : rule == ORBHIP_PROJ_RATIO ? RULE_RATIO : RULE_BEST,
                         nnratio, th_dist, check_orientation, d_match_of_feature, d_nmatches)))
        return rc;
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(m->ev1, m->stream));
    return ORBHIP_OK;
}

int orbhip_best_in_radius_device(orbhip_matcher* m, const OrbProbe* d_probes, const uint8_t* d_probe_desc,
                                 const int32_t* d_nprobes, int probe_cap, const OrbKeypoint* d_kp2, const uint8_t* d_desc2,
                                 const int32_t* d_cell_start2, const int32_t* d_cell_items2, int batch, int cap,
                                 OrbFrameBounds bounds, int32_t* d_best_idx, int32_t* d_best_dist) {
    int rc = check_frames(m, batch, cap, bounds);
    if (rc) return rc;
    if (probe_cap < 1 || probe_cap > (1 << 22)) return orb_fail(ORBHIP_ECAP, "probe_cap %d outside 1..%d", probe_cap, 1 << 22);
    if (batch == 0) return ORBHIP_OK;
    if (!d_probes || !d_probe_desc || !d_nprobes || !d_kp2 || !d_desc2 || !d_cell_start2 || !d_cell_items2 || !d_best_idx ||
        !d_best_dist)
        return orb_fail(ORBHIP_EINVAL, "null buffer");
    HIPCHK(hipSetDevice(m->device));
    (void)hipGetLastError();
    QuerySource s;
    s.mode = MODE_PROBE;
    s.queries = nullptr;
    s.probes = d_probes;
    s.kp1 = nullptr;
    s.n1 = d_nprobes;
    s.prev = nullptr;
    s.valid1 = nullptr;
    s.r = 0.f;
    s.min_level = s.max_level = -1;
    s.nq = probe_cap;
    HIPCHK(hipEventRecord(m->ev0, m->stream));
    dim3 qgrid((unsigned)(((long long)probe_cap * AREA_LPQ + 255) / 256), batch);
    hipLaunchKernelGGL(k_best_in_radius, qgrid, dim3(256), 0, m->stream, s, (const KpDev*)d_kp2, d_cell_start2, d_cell_items2, cap,
                       make_geom(bounds), d_probe_desc, d_desc2, d_best_idx, d_best_dist);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(m->ev1, m->stream));
    return ORBHIP_OK;
}

}  // extern "C"

namespace {

// Host-pointer forms: one frame pair, copies in and out through a staging arena of the handle.
struct PairStage {
    OrbKeypoint *kp1, *kp2;
    uint8_t *desc1, *desc2, *valid1;
    int32_t *n, *cell_start, *cell_items, *matches, *nmatches;
    float* prev;
};

int stage_pair(orbhip_matcher* m, int cap, PairStage& ps) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += align256(bytes);
        return o;
    };
    const size_t o_kp1 = take(sizeof(OrbKeypoint) * cap), o_kp2 = take(sizeof(OrbKeypoint) * cap);
    const size_t o_d1 = take(32 * (size_t)cap), o_d2 = take(32 * (size_t)cap), o_v = take(cap);
    const size_t o_n = take(sizeof(int32_t) * 2), o_cs = take(sizeof(int32_t) * (GRID_CELLS + 1));
    const size_t o_ci = take(sizeof(int32_t) * cap), o_m = take(sizeof(int32_t) * cap), o_nm = take(sizeof(int32_t));
    const size_t o_prev = take(sizeof(float) * 2 * cap);
    if (off > m->stage_bytes) {
        if (m->d_stage) hipFree(m->d_stage);
        m->d_stage = nullptr;
        m->stage_bytes = 0;
        if (hipMalloc(&m->d_stage, off) != hipSuccess) return orb_fail(ORBHIP_ENOMEM, "hipMalloc of %zu bytes failed", off);
        m->stage_bytes = off;
    }
    char* b = (char*)m->d_stage;
    ps.kp1 = (OrbKeypoint*)(b + o_kp1);
    ps.kp2 = (OrbKeypoint*)(b + o_kp2);
    ps.desc1 = (uint8_t*)(b + o_d1);
    ps.desc2 = (uint8_t*)(b + o_d2);
    ps.valid1 = (uint8_t*)(b + o_v);
    ps.n = (int32_t*)(b + o_n);
    ps.cell_start = (int32_t*)(b + o_cs);
    ps.cell_items = (int32_t*)(b + o_ci);
    ps.matches = (int32_t*)(b + o_m);
    ps.nmatches = (int32_t*)(b + o_nm);
    ps.prev = (float*)(b + o_prev);
    return ORBHIP_OK;
}

int upload_pair(orbhip_matcher* m, PairStage& ps, const OrbKeypoint* kp1, const uint8_t* desc1, int n1, const OrbKeypoint* kp2,
                const uint8_t* desc2, int n2) {
    const int32_t n[2] = {n1, n2};
    HIPCHK(hipMemcpyAsync(ps.n, n, sizeof(n), hipMemcpyHostToDevice, m->stream));
    if (n1) {
        HIPCHK(hipMemcpyAsync(ps.kp1, kp1, sizeof(OrbKeypoint) * n1, hipMemcpyHostToDevice, m->stream));
        HIPCHK(hipMemcpyAsync(ps.desc1, desc1, 32 * (size_t)n1, hipMemcpyHostToDevice, m->stream));
    }
    if (n2) {
        HIPCHK(hipMemcpyAsync(ps.kp2, kp2, sizeof(OrbKeypoint) * n2, hipMemcpyHostToDevice, m->stream));
        HIPCHK(hipMemcpyAsync(ps.desc2, desc2, 32 * (size_t)n2, hipMemcpyHostToDevice, m->stream));
    }
    HIPCHK(hipStreamSynchronize(m->stream));  // n[] lives on this stack frame
    return ORBHIP_OK;
}

}  // namespace

extern "C" {

int orbhip_search_for_initialization(orbhip_matcher* m, const OrbKeypoint* kp1, const uint8_t* desc1, int n1,
                                     const OrbKeypoint* kp2, const uint8_t* desc2, int n2, OrbFrameBounds bounds,
                                     float* prev_matched, int window_size, float nnratio, int check_orientation,
                                     int32_t* matches12, int32_t* nmatches) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (n1 < 0 || n2 < 0) return orb_fail(ORBHIP_EINVAL, "negative size");
    if (!nmatches || (n1 && (!kp1 || !desc1 || !prev_matched || !matches12)) || (n2 && (!kp2 || !desc2)))
        return orb_fail(ORBHIP_EINVAL, "null buffer");
    const int cap = std::max(1, std::max(n1, n2));
    int rc = check_frames(m, 1, cap, bounds);
    if (rc) return rc;
    HIPCHK(hipSetDevice(m->device));
    PairStage ps;
    if ((rc = stage_pair(m, cap, ps))) return rc;
    if ((rc = upload_pair(m, ps, kp1, desc1, n1, kp2, desc2, n2))) return rc;
    if (n1) HIPCHK(hipMemcpy(ps.prev, prev_matched, sizeof(float) * 2 * n1, hipMemcpyHostToDevice));
    if ((rc = orbhip_frame_grid_device(m, ps.kp2, ps.n + 1, 1, cap, bounds, ps.cell_start, ps.cell_items))) return rc;
    if ((rc = orbhip_search_for_initialization_device(m, ps.kp1, ps.desc1, ps.n, ps.kp2, ps.desc2, ps.n + 1, ps.cell_start,
                                                      ps.cell_items, 1, cap, bounds, ps.prev, window_size, nnratio,
                                                      check_orientation, ps.matches, ps.nmatches)))
        return rc;
    HIPCHK(hipStreamSynchronize(m->stream));  // the device forms do not wait when a reservation is in place
    if (n1) {
        HIPCHK(hipMemcpy(matches12, ps.matches, sizeof(int32_t) * n1, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(prev_matched, ps.prev, sizeof(float) * 2 * n1, hipMemcpyDeviceToHost));
    }
    HIPCHK(hipMemcpy(nmatches, ps.nmatches, sizeof(int32_t), hipMemcpyDeviceToHost));
    return ORBHIP_OK;
}

int orbhip_window_search(orbhip_matcher* m, const OrbKeypoint* kp1, const uint8_t* desc1, int n1, const uint8_t* valid1,
                         const OrbKeypoint* kp2, const uint8_t* desc2, int n2, OrbFrameBounds bounds, int window_size,
                         int min_level, int max_level, float nnratio, int check_orientation, int32_t* matches21,
                         int32_t* nmatches) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (n1 < 0 || n2 < 0) return orb_fail(ORBHIP_EINVAL, "negative size");
    if (!nmatches || (n1 && (!kp1 || !desc1 || !valid1)) || (n2 && (!kp2 || !desc2 || !matches21)))
        return orb_fail(ORBHIP_EINVAL, "null buffer");
    const int cap = std::max(1, std::max(n1, n2));
    int rc = check_frames(m, 1, cap, bounds);
    if (rc) return rc;
    HIPCHK(hipSetDevice(m->device));
    PairStage ps;
    if ((rc = stage_pair(m, cap, ps))) return rc;
    if ((rc = upload_pair(m, ps, kp1, desc1, n1, kp2, desc2, n2))) return rc;
    if (n1) HIPCHK(hipMemcpy(ps.valid1, valid1, n1, hipMemcpyHostToDevice));
    if ((rc = orbhip_frame_grid_device(m, ps.kp2, ps.n + 1, 1, cap, bounds, ps.cell_start, ps.cell_items))) return rc;
    if ((rc = orbhip_window_search_device(m, ps.kp1, ps.desc1, ps.n, ps.valid1, ps.kp2, ps.desc2, ps.n + 1, ps.cell_start,
                                          ps.cell_items, 1, cap, bounds, window_size, min_level, max_level, nnratio,
                                          check_orientation, ps.matches, ps.nmatches)))
        return rc;
    HIPCHK(hipStreamSynchronize(m->stream));  // the device forms do not wait when a reservation is in place
    if (n2) HIPCHK(hipMemcpy(matches21, ps.matches, sizeof(int32_t) * n2, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(nmatches, ps.nmatches, sizeof(int32_t), hipMemcpyDeviceToHost));
    return ORBHIP_OK;
}

int orbhip_search_by_projection(orbhip_matcher* m, const OrbProbe* probes, const uint8_t* probe_desc, int nprobes,
                                const OrbKeypoint* kp2, const uint8_t* desc2, int n2, OrbFrameBounds bounds,
                                const uint8_t* taken, int rule, float nnratio, int th_dist, int check_orientation,
                                int32_t* match_of_feature, int32_t* nmatches) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (nprobes < 0 || n2 < 0) return orb_fail(ORBHIP_EINVAL, "negative size");
    if (!nmatches || (nprobes && (!probes || !probe_desc)) || (n2 && (!kp2 || !desc2 || !match_of_feature)))
        return orb_fail(ORBHIP_EINVAL, "null buffer");
    const int cap = std::max(1, n2), pcap = std::max(1, nprobes);
    int rc = check_frames(m, 1, cap, bounds);
    if (rc) return rc;
    if (pcap > ORBHIP_MAX_PROBES) return orb_fail(ORBHIP_ECAP, "%d probes > %d", nprobes, ORBHIP_MAX_PROBES);
    HIPCHK(hipSetDevice(m->device));
    // staging: probes | probe descriptors | keypoints | descriptors | taken | n[2] | grid | matches | nmatches
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += align256(bytes);
        return o;
    };
    const size_t o_pr = take(sizeof(OrbProbe) * pcap), o_pd = take(32 * (size_t)pcap), o_kp = take(sizeof(OrbKeypoint) * cap);
    const size_t o_d = take(32 * (size_t)cap), o_t = take(cap), o_n = take(sizeof(int32_t) * 2);
    const size_t o_cs = take(sizeof(int32_t) * (GRID_CELLS + 1)), o_ci = take(sizeof(int32_t) * cap);
    const size_t o_m = take(sizeof(int32_t) * cap), o_nm = take(sizeof(int32_t));
    if (off > m->stage_bytes) {
        if (m->d_stage) hipFree(m->d_stage);
        m->d_stage = nullptr;
        m->stage_bytes = 0;
        if (hipMalloc(&m->d_stage, off) != hipSuccess) return orb_fail(ORBHIP_ENOMEM, "hipMalloc of %zu bytes failed", off);
        m->stage_bytes = off;
    }
    char* b = (char*)m->d_stage;
    const int32_t n[2] = {nprobes, n2};
    HIPCHK(hipMemcpyAsync(b + o_n, n, sizeof(n), hipMemcpyHostToDevice, m->stream));
    if (nprobes) {
        HIPCHK(hipMemcpyAsync(b + o_pr, probes, sizeof(OrbProbe) * nprobes, hipMemcpyHostToDevice, m->stream));
        HIPCHK(hipMemcpyAsync(b + o_pd, probe_desc, 32 * (size_t)nprobes, hipMemcpyHostToDevice, m->stream));
    }
    if (n2) {
        HIPCHK(hipMemcpyAsync(b + o_kp, kp2, sizeof(OrbKeypoint) * n2, hipMemcpyHostToDevice, m->stream));
        HIPCHK(hipMemcpyAsync(b + o_d, desc2, 32 * (size_t)n2, hipMemcpyHostToDevice, m->stream));
        if (taken) HIPCHK(hipMemcpyAsync(b + o_t, taken, n2, hipMemcpyHostToDevice, m->stream));
    }
    HIPCHK(hipStreamSynchronize(m->stream));  // n[] lives on this stack frame
    if ((rc = orbhip_frame_grid_device(m, (const OrbKeypoint*)(b + o_kp), (const int32_t*)(b + o_n) + 1, 1, cap, bounds,
                                       (int32_t*)(b + o_cs), (int32_t*)(b + o_ci))))
        return rc;
    if ((rc = orbhip_search_by_projection_device(m, (const OrbProbe*)(b + o_pr), (const uint8_t*)(b + o_pd),
                                                 (const int32_t*)(b + o_n), pcap, (const OrbKeypoint*)(b + o_kp),
                                                 (const uint8_t*)(b + o_d), (const int32_t*)(b + o_n) + 1,
                                                 (const int32_t*)(b + o_cs), (const int32_t*)(b + o_ci), 1, cap, bounds,
                                                 taken ? (const uint8_t*)(b + o_t) : nullptr, rule, nnratio, th_dist,
                                                 check_orientation, (int32_t*)(b + o_m), (int32_t*)(b + o_nm))))
        return rc;
    HIPCHK(hipStreamSynchronize(m->stream));  // the device forms do not wait when a reservation is in place
    if (n2) HIPCHK(hipMemcpy(match_of_feature, b + o_m, sizeof(int32_t) * n2, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(nmatches, b + o_nm, sizeof(int32_t), hipMemcpyDeviceToHost));
    return ORBHIP_OK;
}

int orbhip_best_in_radius(orbhip_matcher* m, const OrbProbe* probes, const uint8_t* probe_desc, int nprobes,
                          const OrbKeypoint* kp2, const uint8_t* desc2, int n2, OrbFrameBounds bounds, int32_t* best_idx,
                          int32_t* best_dist) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (nprobes < 0 || n2 < 0) return orb_fail(ORBHIP_EINVAL, "negative size");
    if ((nprobes && (!probes || !probe_desc || !best_idx || !best_dist)) || (n2 && (!kp2 || !desc2)))
        return orb_fail(ORBHIP_EINVAL, "null buffer");
    if (nprobes == 0) return ORBHIP_OK;
    const int cap = std::max(1, n2), pcap = nprobes;
    int rc = check_frames(m, 1, cap, bounds);
    if (rc) return rc;
    if (pcap > (1 << 22)) return orb_fail(ORBHIP_ECAP, "%d probes > %d", nprobes, 1 << 22);
    HIPCHK(hipSetDevice(m->device));
    // staging: probes | probe descriptors | keypoints | descriptors | n[2] | grid | best index | best distance
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += align256(bytes);
        return o;
    };
    const size_t o_pr = take(sizeof(OrbProbe) * pcap), o_pd = take(32 * (size_t)pcap), o_kp = take(sizeof(OrbKeypoint) * cap);
    const size_t o_d = take(32 * (size_t)cap), o_n = take(sizeof(int32_t) * 2);
    const size_t o_cs = take(sizeof(int32_t) * (GRID_CELLS + 1)), o_ci = take(sizeof(int32_t) * cap);
    const size_t o_bi = take(sizeof(int32_t) * pcap), o_bd = take(sizeof(int32_t) * pcap);
    if (off > m->stage_bytes) {
        if (m->d_stage) hipFree(m->d_stage);
        m->d_stage = nullptr;
        m->stage_bytes = 0;
        if (hipMalloc(&m->d_stage, off) != hipSuccess) return orb_fail(ORBHIP_ENOMEM, "hipMalloc of %zu bytes failed", off);
        m->stage_bytes = off;
    }
    char* b = (char*)m->d_stage;
    const int32_t n[2] = {nprobes, n2};
    HIPCHK(hipMemcpyAsync(b + o_n, n, sizeof(n), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(b + o_pr, probes, sizeof(OrbProbe) * nprobes, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(b + o_pd, probe_desc, 32 * (size_t)nprobes, hipMemcpyHostToDevice, m->stream));
    if (n2) {
        HIPCHK(hipMemcpyAsync(b + o_kp, kp2, sizeof(OrbKeypoint) * n2, hipMemcpyHostToDevice, m->stream));
        HIPCHK(hipMemcpyAsync(b + o_d, desc2, 32 * (size_t)n2, hipMemcpyHostToDevice, m->stream));
    }
    HIPCHK(hipStreamSynchronize(m->stream));  // n[] lives on this stack frame
    if ((rc = orbhip_frame_grid_device(m, (const OrbKeypoint*)(b + o_kp), (const int32_t*)(b + o_n) + 1, 1, cap, bounds,
                                       (int32_t*)(b + o_cs), (int32_t*)(b + o_ci))))
        return rc;
    if ((rc = orbhip_best_in_radius_device(m, (const OrbProbe*)(b + o_pr), (const uint8_t*)(b + o_pd), (const int32_t*)(b + o_n),
                                           pcap, (const OrbKeypoint*)(b + o_kp), (const uint8_t*)(b + o_d),
                                           (const int32_t*)(b + o_cs), (const int32_t*)(b + o_ci), 1, cap, bounds,
                                           (int32_t*)(b + o_bi), (int32_t*)(b + o_bd))))
        return rc;
    HIPCHK(hipMemcpyAsync(best_idx, b + o_bi, sizeof(int32_t) * nprobes, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipMemcpyAsync(best_dist, b + o_bd, sizeof(int32_t) * nprobes, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return ORBHIP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:221-250), one wave per map point.
// The N x N distance matrix lives in LDS (u16); lane i then finds the median of row i — the
// element of rank (size_t)(0.5 * (N - 1)) — by bisection on the value (distances are 0..256), and
// the wave takes the first row attaining the smallest median.
namespace {

#define DD_MAX_N 128

__global__ __launch_bounds__(64) void k_distinctive(const uint8_t* __restrict__ desc, const int32_t* __restrict__ start,
                                                    int npoints, int32_t* __restrict__ best_idx,
                                                    int32_t* __restrict__ best_median, int* __restrict__ err) {
    __shared__ uint16_t dist[DD_MAX_N * DD_MAX_N];
    __shared__ uint32_t rows[DD_MAX_N * 8];
    const int p = blockIdx.x, lane = threadIdx.x;
    const int s0 = start[p];
    int N = start[p + 1] - s0;
    if (N <= 0) {
        if (lane == 0) {
            best_idx[p] = -1;
            best_median[p] = INT_MAX;
        }
        return;
    }
    if (N > DD_MAX_N) {  // not representable in the LDS matrix: flagged, nothing is written
        if (lane == 0) {
            atomicOr(err, 8);
            best_idx[p] = -1;
            best_median[p] = INT_MAX;
        }
        return;
    }
    const uint32_t* src = (const uint32_t*)(desc + (long long)s0 * 32);
    for (int i = lane; i < N * 8; i += 64) rows[i] = src[i];
    wave_fence();
    for (int e = lane; e < N * N; e += 64) {
        const int i = e / N, j = e - i * N;
        int d = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) d += __popc(rows[i * 8 + w] ^ rows[j * 8 + w]);
        dist[e] = (uint16_t)d;
    }
    wave_fence();
    const int rank = (int)(0.5 * (N - 1));
    unsigned long long bestkey = KEY_NONE;
    for (int i = lane; i < N; i += 64) {
        // smallest v with #{j : dist[i][j] <= v} > rank
        int lo = 0, hi = 256;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int c = 0;
            for (int j = 0; j < N; j++) c += dist[i * N + j] <= mid;
            if (c > rank) hi = mid;
            else lo = mid + 1;
        }
        bestkey = min(bestkey, ((unsigned long long)lo << 32) | (unsigned)i);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bestkey = min(bestkey, __shfl_xor(bestkey, o));
    if (lane == 0) {
        best_idx[p] = (int32_t)(bestkey & 0xffffffffu);
        best_median[p] = (int32_t)(bestkey >> 32);
    }
}

}  // namespace

extern "C" int orbhip_distinctive_descriptors_device(orbhip_matcher* m, const uint8_t* d_desc, const int32_t* d_start,
                                                     int npoints, int32_t* d_best_idx, int32_t* d_best_median) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (npoints < 0) return orb_fail(ORBHIP_EINVAL, "negative size");
    if (npoints == 0) return ORBHIP_OK;
    if (!d_desc || !d_start || !d_best_idx || !d_best_median) return orb_fail(ORBHIP_EINVAL, "null buffer");
    HIPCHK(hipSetDevice(m->device));
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_distinctive, dim3(npoints), dim3(64), 0, m->stream, d_desc, d_start, npoints, d_best_idx,
                       d_best_median, m->d_err);
    HIPCHK(hipGetLastError());
    return ORBHIP_OK;
}
