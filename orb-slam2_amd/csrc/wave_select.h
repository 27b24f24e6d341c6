// Wave-cooperative form of the selection in introselect.h: the same element movement as
// libstdc++'s std::nth_element (response-greater comparator on the response bits of the keys),
// with the Hoare partition of each introselect round evaluated by the 64 lanes of one wave.
//
// Why the partition can be done in parallel and still move elements exactly like the serial scan:
// with pivot score ps, call position i a left stop if score(v[i]) <= ps (the left scan halts there)
// and a right stop if score(v[i]) >= ps.  Let L[0] < L[1] < ... be the left stops of the range in
// ascending order and R[0] > R[1] > ... the right stops in descending order, both taken on the
// array as it is before the round.  The serial loop swaps exactly the pairs (L[k], R[k]) with
// L[k] < R[k] — the first K pairs, since L ascends and R descends — because the region between
// the two scan pointers is always unmodified, and returns cut = L[0] if K == 0, else
// min(L[K], R[K-1]) (the left scan of the last round halts at the next original left stop or, at
// the latest, at R[K-1], which now holds the element swapped in from L[K-1]).
//
// v, lpos and rpos live in LDS and belong to the calling wave; every lane of the wave must call
// with the same arguments.
#ifndef ORBHIP_WAVE_SELECT_H
#define ORBHIP_WAVE_SELECT_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "introselect.h"

namespace orbsel {

__device__ __forceinline__ int lane_prefix64(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
// the key a given lane holds, to every lane (src is wave-uniform)
__device__ __forceinline__ uint32_t lane_value(uint32_t x, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)x, src); }
__device__ __forceinline__ uint64_t lane_value(uint64_t x, int src) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(x >> 32), src) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, src);
}

template <class KeyT>
__device__ inline void wave_nth_element_desc(KeyT* v, int n, int nth, uint16_t* lpos, uint16_t* rpos, int lane,
                                             int depth_limit = -1) {
    if (n == 0 || nth == n) return;
    int first = 0, last = n;
    int depth = depth_limit < 0 ? 2 * floor_log2(n) : depth_limit;
    while (last - first > 3) {
        if (depth == 0) {
            if (lane == 0) {
                heap_select(v, first, nth + 1, last);
                swap_at(v, first, nth);
            }
            wave_lds_fence();
            return;
        }
        --depth;
        const int mid = first + (last - first) / 2;
        // median of v[first + 1], v[mid], v[last - 1] to v[first] (introselect.h: median_to): the four keys are read
        // by four lanes at once and the decision is made on their broadcast values, so the pivot is known without
        // another trip to LDS
        const int p4 = lane == 0 ? first : lane == 1 ? first + 1 : lane == 2 ? mid : last - 1;
        KeyT c4 = 0;
        if (lane < 4) c4 = v[p4];
        const KeyT vd = lane_value(c4, 0), va = lane_value(c4, 1), vb = lane_value(c4, 2), vc = lane_value(c4, 3);
        int pm;
        KeyT vm;
        if (score_gt(va, vb)) {
            if (score_gt(vb, vc)) pm = mid, vm = vb;
            else if (score_gt(va, vc)) pm = last - 1, vm = vc;
            else pm = first + 1, vm = va;
        } else if (score_gt(va, vc)) pm = first + 1, vm = va;
        else if (score_gt(vb, vc)) pm = last - 1, vm = vc;
        else pm = mid, vm = vb;
        if (lane == 0) {
            v[first] = vm;
            v[pm] = vd;
        }
        wave_lds_fence();
        const uint32_t ps = score_of(vm);
        const int lo = first + 1, hi = last;
        int nL = 0, nR = 0;
        for (int base = lo; base < hi; base += 64) {
            const int i = base + lane;
            const bool in = i < hi;
            const uint32_t sc = in ? score_of(v[i]) : 0u;
            const bool isL = in && !(sc > ps);
            const bool isR = in && !(ps > sc);
            const unsigned long long mL = __ballot(isL), mR = __ballot(isR);
            if (isL) lpos[nL + lane_prefix64(mL)] = (uint16_t)i;
            if (isR) rpos[nR + lane_prefix64(mR)] = (uint16_t)i;  // ascending; R[k] = rpos[nR - 1 - k]
            nL += __popcll(mL);
            nR += __popcll(mR);
        }
        wave_lds_fence();
        const int mn = min(nL, nR);
        // the first 64 (L[k], R[k]) pairs stay in registers: they decide K, are the pairs to swap and give the cut
        int la = 0, rb = 0;
        if (lane < nL) la = lpos[lane];
        if (lane < nR) rb = rpos[nR - 1 - lane];
        int K = __popcll(__ballot(lane < mn && la < rb));
        if (K == 64) {
            for (int base = 64; base < mn; base += 64) {
                const int k = base + lane;
                const bool sw = k < mn && lpos[k] < rpos[nR - 1 - k];
                const int c = __popcll(__ballot(sw));
                K += c;
                if (c < 64) break;
            }
        }
        if (lane < K) {
            const KeyT ta = v[la], tb = v[rb];
            v[la] = tb;
            v[rb] = ta;
        }
        for (int k = 64 + lane; k < K; k += 64) {
            const int a = lpos[k], b = rpos[nR - 1 - k];
            const KeyT ta = v[a], tb = v[b];
            v[a] = tb;
            v[b] = ta;
        }
        int cut;
        if (K == 0) {
            cut = __builtin_amdgcn_readlane(la, 0);  // L[0]; nL >= 1: the scan always stops at the sentinel
        } else {
            const int rk = K <= 64 ? __builtin_amdgcn_readlane(rb, K - 1) : (int)rpos[nR - K];
            const int lk = K < nL ? (K < 64 ? __builtin_amdgcn_readlane(la, K) : (int)lpos[K]) : rk;
            cut = min(lk, rk);
        }
        cut = __builtin_amdgcn_readfirstlane(cut);
        wave_lds_fence();
        if (cut <= nth) first = cut;
        else last = cut;
    }
    if (lane == 0) insertion_sort(v, first, last);
    wave_lds_fence();
}

}  // namespace orbsel
#endif
