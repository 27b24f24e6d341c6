// The Hamming k=2 scan on the matrix cores of gfx950.
//
// With every descriptor bit b mapped to a signed byte (train: b ? +1 : -1, query: b ? -1 : +1) the dot product of
// a train row and a query is (#differing bits) - (#equal bits) = 2 d - 256, an exact small integer, and
// v_mfma_i32_16x16x64_i8 forms 16 x 16 of them over 64 bits in one instruction: four instructions give 2 d - 256
// for 16 train rows x 16 queries.  That is 4 distances per SIMD cycle against the
// ~1.2 per cycle the xor + popcount instruction stream can issue (knn2_kernel.h), and the result is the same
// integer.  The two operands are laid out with the same bit -> (lane, byte) rule, so the order in which the
// instruction walks its 64 products is irrelevant.
//
// One workgroup = 256 queries (four waves x four 16-query tiles whose expanded bits stay in registers for the
// whole scan) against one chunk of train rows, with the arguments, the keys (distance << 22 | chunk-relative
// row) and the partial results of k_knn2, so the two kernels are interchangeable.  Train rows are read as
// bits (32 bytes per row), expanded through a 256-entry byte -> 8 bytes table in LDS into a double-buffered
// stage of 32 rows, and read back as operand fragments (row pitch 80 bytes inside a 16-row x 64-bit block:
// the sixteen rows a ds_read_b128 touches fall into distinct banks).
//
// A lane holds, per query tile, the results of rows 4 (lane / 16) .. + 3 of each 16-row tile for query lane % 16,
// so it keeps the two smallest keys over a quarter of the rows; the four lanes of a query are merged at the end.
// Rows ascend within a lane, so a key can only enter the lane's two smallest if its distance is below the
// lane's second-best distance: one wave vote per query tile on the minimum of its four results skips the
// keyed folds (which are exact whatever the vote lets through).
#pragma once
#include <type_traits>

#include "knn2_kernel.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));

#define KMM_STAGE_ROWS 32
#define KMM_ROW_PITCH 80                                        // 64 operand bytes of a k-step + 16
#define KMM_BLOCK_BYTES (16 * KMM_ROW_PITCH)                    // 16 rows of one k-step
#define KMM_STAGE_BYTES ((KMM_STAGE_ROWS / 16) * 4 * KMM_BLOCK_BYTES)
#define KMM_TABLE_BYTES 2048
#define KMM_THR_NONE 0x400                                      // above every doubled distance

// nibble n -> four bytes, byte j = bit j of n ? 0x01 : 0xff
__device__ __forceinline__ uint32_t kmm_nibble(uint32_t n) {
    const uint32_t s = (n * 0x00204081u) & 0x01010101u;  // byte j = bit j (the four shifted copies do not overlap)
    return 0xffffffffu - s * 0xfeu;
}

__global__ __launch_bounds__(256) void k_knn2_mm(const uint32_t* __restrict__ q, int nq, long long q_pair_words,
                                                 const uint32_t* __restrict__ t, int nt, long long t_pair_words,
                                                 int chunk_rows, int nchunks, int idx_base, uint2* __restrict__ part,
                                                 int32_t* __restrict__ o_idx, int32_t* __restrict__ o_d1,
                                                 int32_t* __restrict__ o_d2) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[KMM_TABLE_BYTES + 2 * KMM_STAGE_BYTES];
    uint2* table = (uint2*)lds;
    uint8_t* stages = lds + KMM_TABLE_BYTES;
    const int pair = blockIdx.z, chunk = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, col = lane & 15;
    table[tid] = make_uint2(kmm_nibble((uint32_t)tid & 15u), kmm_nibble((uint32_t)tid >> 4));
    __syncthreads();

    // ---- the lane's share of the four query tiles: bits 64 s + 16 g .. + 15 of query `col`, k-step s ----
    i32x4 bq[4][4];
#pragma unroll
    for (int qt = 0; qt < 4; qt++) {
        const int qi = blockIdx.x * 256 + wave * 64 + qt * 16 + col;
        const uint32_t* qp = q + (long long)pair * q_pair_words + (long long)min(qi, nq - 1) * 8;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const uint32_t bits = ~(qp[2 * s + (g >> 1)] >> (16 * (g & 1)));  // complemented: query bytes are -train bytes
            const uint2 lo = table[bits & 0xffu], hi = table[(bits >> 8) & 0xffu];
            bq[qt][s] = i32x4{(int)lo.x, (int)lo.y, (int)hi.x, (int)hi.y};
        }
    }

    const int j0 = chunk * chunk_rows;
    const int rows = min(nt - j0, chunk_rows);
    const uint32_t* __restrict__ tp = t + (long long)pair * t_pair_words + (long long)j0 * 8;
    uint32_t best[4], best2[4];
    int thr[4];
#pragma unroll
    for (int qt = 0; qt < 4; qt++) {
        best[qt] = best2[qt] = KNN_NONE;
        thr[qt] = KMM_THR_NONE;
    }

    // expansion: thread = (row of the stage, word of the row); word w holds bits 32 w .. + 31 = k-step w / 2,
    // operand bytes 32 (w & 1) .. + 31 of the row
    const int xrow = tid >> 3, xw = tid & 7;
    const int xoff = (((xrow >> 4) * 4 + (xw >> 1)) * 16 + (xrow & 15)) * KMM_ROW_PITCH + 32 * (xw & 1);
    const int aoff = col * KMM_ROW_PITCH + 16 * g;  // fragment of the lane inside a (tile, k-step) block
    const int nstages = (rows + KMM_STAGE_ROWS - 1) / KMM_STAGE_ROWS;
    uint32_t pre = 0;
    if (rows > 0) pre = tp[(long long)min(xrow, rows - 1) * 8 + xw];
    // One stage: expand the 32 rows whose bits are in `pre`, fetch the next stage's bits, multiply.  TAIL = some
    // rows of the stage lie past the chunk (their keys are replaced by "none").
    auto stage = [&](const int st, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        uint8_t* buf = stages + (st & 1) * KMM_STAGE_BYTES;
        {
            const uint2 e0 = table[pre & 0xffu], e1 = table[(pre >> 8) & 0xffu], e2 = table[(pre >> 16) & 0xffu],
                        e3 = table[pre >> 24];
            *(uint4*)(buf + xoff) = make_uint4(e0.x, e0.y, e1.x, e1.y);
            *(uint4*)(buf + xoff + 16) = make_uint4(e2.x, e2.y, e3.x, e3.y);
        }
        // the next stage's bits travel while this one is multiplied (rows past the chunk re-read its last row;
        // their results are never folded)
        if (st + 1 < nstages) pre = tp[(long long)min((st + 1) * KMM_STAGE_ROWS + xrow, rows - 1) * 8 + xw];
        __syncthreads();
#pragma unroll
        for (int tile = 0; tile < KMM_STAGE_ROWS / 16; tile++) {
            i32x4 a[4];
#pragma unroll
            for (int s = 0; s < 4; s++) a[s] = *(const i32x4*)(buf + (tile * 4 + s) * KMM_BLOCK_BYTES + aoff);
            const int rbase = st * KMM_STAGE_ROWS + tile * 16 + 4 * g;
            // key = (2 d << 21) | row with 2 d = acc + 256: one shift-add per result
            const uint32_t kbase = (256u << (KNN_IDX_BITS - 1)) + (uint32_t)rbase;
#pragma unroll
            for (int qt = 0; qt < 4; qt++) {
                // accumulator input 0 (an inline constant, no registers to preset): acc[r] = 2 d - 256
                i32x4 acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[0], bq[qt][0], i32x4{0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
                for (int s = 1; s < 4; s++) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s], bq[qt][s], acc, 0, 0, 0);
                const int m = min(min(acc[0], acc[1]), min(acc[2], acc[3]));
                if (__builtin_amdgcn_ballot_w64(m < thr[qt]) != 0) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        uint32_t key = ((uint32_t)acc[r] << (KNN_IDX_BITS - 1)) + (kbase + (uint32_t)r);
                        if (TAIL && rbase + r >= rows) key = KNN_NONE;
                        fold_key(key, best[qt], best2[qt]);
                    }
                    // second-best 2 d - 256 (a row index above 2^21 adds one: the vote then lets an equal distance
                    // through, which the fold rejects)
                    thr[qt] = best2[qt] == KNN_NONE ? KMM_THR_NONE : (int)(best2[qt] >> (KNN_IDX_BITS - 1)) - 256;
                }
            }
        }
    };
    const int nfull = rows / KMM_STAGE_ROWS;
    for (int st = 0; st < nfull; st++) stage(st, std::false_type{});
    if (nfull < nstages) stage(nfull, std::true_type{});

    // ---- the four lanes of a query (same lane % 16) merge their two smallest keys ----
#pragma unroll
    for (int qt = 0; qt < 4; qt++) {
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const uint32_t ob = (uint32_t)__shfl_xor((int)best[qt], off), ob2 = (uint32_t)__shfl_xor((int)best2[qt], off);
            fold_key(ob, best[qt], best2[qt]);
            fold_key(ob2, best[qt], best2[qt]);
        }
        const int qi = blockIdx.x * 256 + wave * 64 + qt * 16 + col;
        if (g != 0 || qi >= nq) continue;
        const uint32_t b = best[qt], b2 = best2[qt];
        if (nchunks == 1) {
            const long long o = (long long)pair * nq + qi;
            o_idx[o] = b == KNN_NONE ? -1 : (int)(b & KNN_IDX_MASK) + idx_base;
            o_d1[o] = b == KNN_NONE ? INT_MAX : (int)(b >> KNN_IDX_BITS);
            o_d2[o] = b2 == KNN_NONE ? INT_MAX : (int)(b2 >> KNN_IDX_BITS);
        } else {
            part[((long long)pair * nchunks + chunk) * nq + qi] = make_uint2(b, b2);
        }
    }
}
