// The ordered merge of per-shard (idx, d1, d2) records of index-contiguous train shards, one source for the
// device kernel (matcher.hip: k_knn2_merge) and its host twin (common.cpp: orbhip_knn2_merge_host).
//
// Shards are visited in ascending order, i.e. in ascending global row order, with the update rule of the
// reference's scan (src/ORBmatcher.cc:461-485): a shard's best enters like one row, its second-best like another
// one behind it, so the result is the serial ascending scan's — the first row attaining the minimum wins, the
// second-best counts multiplicity.  parts: [nshards][3][nq] int32 (idx, d1, d2 per shard); idx < 0 marks an
// empty shard.
#pragma once
#include <climits>
#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#define ORB_HD __host__ __device__
#else
#define ORB_HD
#endif

ORB_HD inline void knn2_merge_query(const int32_t* parts, int nshards, int nq, int qi, int32_t* o_idx, int32_t* o_d1,
                                    int32_t* o_d2) {
    int best = INT_MAX, best2 = INT_MAX, idx = -1;
    for (int s = 0; s < nshards; s++) {
        const int32_t* p = parts + (long long)s * 3 * nq;
        const int i = p[qi], d1 = p[nq + qi], d2 = p[2 * nq + qi];
        if (i < 0) continue;
        if (d1 < best) {
            best2 = best;
            best = d1;
            idx = i;
        } else if (d1 < best2) {
            best2 = d1;
        }
        if (d2 < best2) best2 = d2;
    }
    o_idx[qi] = idx;
    o_d1[qi] = best;
    o_d2[qi] = best2;
}

// Layout of the all-gather's receive buffer: rank r's record of 3 * nq int32 starts at int32 offset 3 * nq * r.
ORB_HD inline size_t knn2_record_ints(int nq) { return (size_t)3 * (size_t)nq; }
ORB_HD inline size_t knn2_slot_offset_ints(int nq, int rank) { return knn2_record_ints(nq) * (size_t)rank; }
