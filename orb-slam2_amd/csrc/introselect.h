// Selection of the n best keypoints with the exact element movement of libstdc++'s
// std::nth_element (GCC 11 bits/stl_algo.h __introselect and the helpers it calls), because the
// reference keeps "the first n elements after nth_element" of each cell and level
// (src/ORBextractor.cc:683-685,697-701 through cv::KeyPointsFilter::retainBest), so both which
// of several equal-score keypoints survive and the order they come out in are defined by that
// algorithm's swap sequence.  FAST scores are small integers and ties are the norm.
//
// Elements are keys whose top bits hold the response and whose low bits hold the payload
// (coordinates); the comparator is the reference's "response greater" and looks at the response
// only.  Two key formats:
//   uint32_t  FAST score (an integer < 256) in bits 31..24;
//   uint64_t  any float response, mapped to an unsigned integer of the same order, in bits 63..32
//             (HARRIS_SCORE, src/ORBextractor.cc:616-620).
#ifndef ORBHIP_INTROSELECT_H
#define ORBHIP_INTROSELECT_H
#include <stdint.h>
#include "hd.h"

namespace orbsel {

ORB_HD inline uint32_t score_of(uint32_t k) { return k >> 24; }
ORB_HD inline uint32_t score_of(uint64_t k) { return (uint32_t)(k >> 32); }
ORB_HD inline bool score_gt(uint32_t a, uint32_t b) { return score_of(a) > score_of(b); }
ORB_HD inline bool score_gt(uint64_t a, uint64_t b) { return score_of(a) > score_of(b); }

// element type behind a pointer-like P
template <class P> struct elem_of;
template <class T> struct elem_of<T*> { typedef T type; };

template <class P>
ORB_HD inline void swap_at(P v, int i, int j) {
    typename elem_of<P>::type t = v[i];
    v[i] = v[j];
    v[j] = t;
}

// Median of v[a], v[b], v[c] under score_gt goes to v[dst].
template <class P>
ORB_HD inline void median_to(P v, int dst, int a, int b, int c) {
    typename elem_of<P>::type va = v[a], vb = v[b], vc = v[c];
    int m;
    if (score_gt(va, vb)) {
        if (score_gt(vb, vc)) m = b;
        else if (score_gt(va, vc)) m = c;
        else m = a;
    } else if (score_gt(va, vc)) m = a;
    else if (score_gt(vb, vc)) m = c;
    else m = b;
    swap_at(v, dst, m);
}

// Hoare partition of [first, last) around pivot (a value, held outside the range); both scans
// are unguarded and rely on the median-of-three sentinels.
template <class P>
ORB_HD inline int partition_unguarded(P v, int first, int last, typename elem_of<P>::type pivot) {
    for (;;) {
        while (score_gt(v[first], pivot)) ++first;
        --last;
        while (score_gt(pivot, v[last])) --last;
        if (!(first < last)) return first;
        swap_at(v, first, last);
        ++first;
    }
}

// Sift `value` down from `hole` in the heap v[base .. base+len), then back up (libstdc++'s
// __adjust_heap followed by its __push_heap).
template <class P>
ORB_HD inline void heap_adjust(P v, int base, int hole, int len, typename elem_of<P>::type value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (score_gt(v[base + child], v[base + child - 1])) child--;
        v[base + hole] = v[base + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        v[base + hole] = v[base + child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && score_gt(v[base + parent], value)) {
        v[base + hole] = v[base + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    v[base + hole] = value;
}

template <class P>
ORB_HD inline void heap_select(P v, int first, int middle, int last) {
    const int len = middle - first;
    if (len >= 2) {
        int parent = (len - 2) / 2;
        for (;;) {
            typename elem_of<P>::type value = v[first + parent];
            heap_adjust(v, first, parent, len, value);
            if (parent == 0) break;
            parent--;
        }
    }
    for (int i = middle; i < last; ++i)
        if (score_gt(v[i], v[first])) {
            typename elem_of<P>::type value = v[i];
            v[i] = v[first];
            heap_adjust(v, first, 0, len, value);
        }
}

template <class P>
ORB_HD inline void insertion_sort(P v, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        typename elem_of<P>::type val = v[i];
        if (score_gt(val, v[first])) {
            for (int k = i; k > first; --k) v[k] = v[k - 1];
            v[first] = val;
        } else {
            int hole = i, next = i - 1;
            while (score_gt(val, v[next])) {
                v[hole] = v[next];
                hole = next;
                --next;
            }
            v[hole] = val;
        }
    }
}

ORB_HD inline int floor_log2(int n) {
    int l = 0;
    while (n > 1) {
        n >>= 1;
        ++l;
    }
    return l;
}

// Rearranges v[0 .. n) exactly as std::nth_element(v, v + nth, v + n, response-greater) does.
// depth_limit < 0 means the library's own 2*floor(log2(n)); tests pass small limits to reach the
// heap-select branch.
template <class P>
ORB_HD inline void nth_element_desc(P v, int n, int nth, int depth_limit = -1) {
    if (n == 0 || nth == n) return;
    int first = 0, last = n;
    int depth = depth_limit < 0 ? 2 * floor_log2(n) : depth_limit;
    while (last - first > 3) {
        if (depth == 0) {
            heap_select(v, first, nth + 1, last);
            swap_at(v, first, nth);
            return;
        }
        --depth;
        int mid = first + (last - first) / 2;
        median_to(v, first, first + 1, mid, last - 1);
        int cut = partition_unguarded(v, first + 1, last, v[first]);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    insertion_sort(v, first, last);
}

}  // namespace orbsel
#endif
