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
        if (lane == 0) median_to(v, first, first + 1, mid, last - 1);
        wave_lds_fence();
        const uint32_t ps = score_of(v[first]);
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
        int K = 0;
        for (int base = 0; base < mn; base += 64) {
            const int k = base + lane;
            const bool sw = k < mn && lpos[k] < rpos[nR - 1 - k];
            const int c = __popcll(__ballot(sw));
            K += c;
            if (c < 64) break;
        }
        for (int k = lane; k < K; k += 64) {
            const int a = lpos[k], b = rpos[nR - 1 - k];
            const KeyT ta = v[a], tb = v[b];
            v[a] = tb;
            v[b] = ta;
        }
        int cut;
        if (K == 0) cut = lpos[0];
        else {
            const int rk = rpos[nR - K];
            cut = K < nL ? min((int)lpos[K], rk) : rk;
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
