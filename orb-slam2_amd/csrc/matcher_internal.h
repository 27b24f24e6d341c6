// State of a matcher handle, shared by the translation units that implement its entry points.
#ifndef ORBHIP_MATCHER_INTERNAL_H
#define ORBHIP_MATCHER_INTERNAL_H
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "../../include/orbhip.h"
#include "common.h"

#define HIPCHK(call)                                                                         \
    do {                                                                                     \
        hipError_t err__ = (call);                                                           \
        if (err__ != hipSuccess)                                                             \
            return orb_fail(ORBHIP_EHIP, "%s failed: %s", #call, hipGetErrorString(err__)); \
    } while (0)

struct orbhip_matcher {
    int device;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.f;
    void* d_part = nullptr;
    size_t part_bytes = 0;
    void* d_scratch = nullptr;  // this device's fixed scratch array (not owned; never freed)
    void* d_gather = nullptr;  // receive buffer of the map-sharded scan's all-gather
    size_t gather_bytes = 0;
    void* d_stage = nullptr;  // staging of the host-pointer entry point
    size_t stage_bytes = 0;
    void* d_work = nullptr;  // workspace of the windowed matchers
    size_t work_bytes = 0;
    void* d_topk = nullptr;  // per probe: its four best candidates and its candidate count (projection matchers)
    size_t topk_bytes = 0;
    int* d_err = nullptr;  // device error flags (int) + the candidate count a failed reservation needed (int64 at +8), read by orbhip_matcher_synchronize
    int scan_kernel = 0;  // ORBHIP_SCAN_MATRIX / ORBHIP_SCAN_POPCOUNT: which kernel scans the train rows of orbhip_knn2*
    long long cand_reserved = 0;  // > 0: candidate lists of the windowed / projection matchers live in a reservation (asynchronous form)
    int sm_count = 256;
};

// The matrix-core scan kernel lives in a translation unit of its own (knn2_matrix.hip): it is compiled with
// floating-point options (IEEE mode off, no NaNs) that must not reach any other kernel.  Queues the kernel on
// `stream`; grid (qblocks, nchunks, npairs), arguments as k_knn2's.
void orb_launch_knn2_matrix(hipStream_t stream, int qblocks, int nchunks, int npairs, const uint32_t* d_q, int nq,
                            long long q_pair_words, const uint32_t* d_t, int nt, long long t_pair_words, int chunk_rows,
                            int idx_base, uint2* d_part, int32_t* d_idx, int32_t* d_d1, int32_t* d_d2);

#endif
