// Translation unit of the matrix-core Hamming scan (knn2_mfma.h) alone.
//
// The kernel's float keys are never NaN, and its matrix results feed v_min_f32 / v_max_f32 directly, so this file
// is compiled with IEEE mode off and NaNs not honoured (in IEEE mode every matrix result would first be quieted by
// a v_max_f32 x, x of its own: 32 more instructions per tile), and with matrix results in vector registers.  Those
// options change code generation of every kernel they touch — the xor + popcount scan, compiled with them, keeps its
// train rows in vector instead of scalar registers and runs at half its rate — which is why nothing else is
// compiled here.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "matcher_internal.h"

namespace {
#define KNN_HELPERS_ONLY
#include "knn2_mfma.h"
}  // namespace

void orb_launch_knn2_matrix(hipStream_t stream, int qblocks, int nchunks, int npairs, const uint32_t* d_q, int nq,
                            long long q_pair_words, const uint32_t* d_t, int nt, long long t_pair_words, int chunk_rows,
                            int idx_base, uint2* d_part, int32_t* d_idx, int32_t* d_d1, int32_t* d_d2) {
    // tiles of a chunk that go through the branch-free insertion before the vote path takes over
    static const int sort_tiles = [] {
        const char* e = getenv("ORBHIP_KNN_SORT_TILES");  // tuning experiments only
        const int v = e ? atoi(e) : KMM_SORT_TILES_DEFAULT;
        return std::max(0, std::min(v, (int)KMM_SORT_TILES_MAX));
    }();
    hipLaunchKernelGGL(k_knn2_mm, dim3(qblocks, nchunks, npairs), dim3(256), 0, stream, d_q, nq, q_pair_words, d_t, nt,
                       t_pair_words, chunk_rows, nchunks, idx_base, d_part, d_idx, d_d1, d_d2, sort_tiles);
}
