#include "common.h"

#include <cstdarg>
#include <cstdio>

#include "../../include/orbhip.h"
#include "knn2_merge.h"

static thread_local char g_last_error[512] = "";

int orb_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" {

const char* orbhip_last_error(void) { return g_last_error; }
const char* orbhip_version(void) { return "orbhip 0.1 (gfx950)"; }

// ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1805-1821): the reference's SWAR bit count
// over eight 32-bit words equals the population count of the xor.
int orbhip_hamming_distance(const void* a, const void* b) {
    const unsigned long long* pa = (const unsigned long long*)a;
    const unsigned long long* pb = (const unsigned long long*)b;
    unsigned long long wa[4], wb[4];
    __builtin_memcpy(wa, pa, 32);
    __builtin_memcpy(wb, pb, 32);
    return __builtin_popcountll(wa[0] ^ wb[0]) + __builtin_popcountll(wa[1] ^ wb[1]) +
           __builtin_popcountll(wa[2] ^ wb[2]) + __builtin_popcountll(wa[3] ^ wb[3]);
}

// Host twin of orbhip_knn2_merge_device: the same function (knn2_merge.h) over records that lie in host memory,
// for a host that exchanges the per-shard records through a transport of its own (MPI, gloo, a socket).
int orbhip_knn2_merge_host(const int32_t* parts, int nshards, int nq, int32_t* idx, int32_t* d1, int32_t* d2) {
    if (nshards < 1 || nq < 0) return orb_fail(ORBHIP_EINVAL, "bad size");
    if (nq == 0) return ORBHIP_OK;
    if (!parts || !idx || !d1 || !d2) return orb_fail(ORBHIP_EINVAL, "null buffer");
    for (int qi = 0; qi < nq; qi++) knn2_merge_query(parts, nshards, nq, qi, idx, d1, d2);
    return ORBHIP_OK;
}

int orbhip_knn2_gather_layout(int nq, int world, int rank, size_t* record_ints, size_t* slot_offset_ints, size_t* total_bytes) {
    if (nq < 0 || world < 1 || rank < 0 || rank >= world) return orb_fail(ORBHIP_EINVAL, "rank %d of %d, %d queries", rank, world, nq);
    if (record_ints) *record_ints = knn2_record_ints(nq);
    if (slot_offset_ints) *slot_offset_ints = knn2_slot_offset_ints(nq, rank);
    if (total_bytes) *total_bytes = knn2_record_ints(nq) * (size_t)world * sizeof(int32_t);
    return ORBHIP_OK;
}

}  // extern "C"
