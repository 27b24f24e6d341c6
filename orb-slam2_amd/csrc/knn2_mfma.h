// The Hamming k=2 scan on the matrix cores of gfx950.
//
// With the bits t_k of a train row as bytes 0 / 1 and the bits q_k of a query as bytes 1 - 2 q_k (+1 / -1), the dot
// product sum t_k (1 - 2 q_k) = popcount(t) - 2 popcount(t & q) is the Hamming distance minus popcount(q): an exact
// small integer that orders the rows of one query like the distance does.  v_mfma_i32_32x32x32_i8 forms 32 x 32
// such products over 32 bits in one instruction, so eight instructions give d - popcount(q) for 32 train rows x 32
// queries: 4 distances per SIMD cycle at the instruction's issue rate, against the ~1.2 per cycle the xor +
// popcount instruction stream can issue (knn2_kernel.h), and the result is the same integer.  The order in which
// the instruction walks its products is irrelevant as long as both operands use the same bit -> (lane, byte)
// rule; the rule here is the one that is cheapest to produce: the lanes of half h = lane / 32 own words 4 h .. 4 h + 3
// of a row; word i feeds instructions 2 i and 2 i + 1, and byte j of dword k of its eight operand dwords is bit
// k + 8 j of the word, i.e. (word >> k) & 0x01010101.
//
// One workgroup = 256 queries (four waves x two 32-query tiles whose expanded bits stay in registers for the
// whole scan) against one chunk of train rows, with the arguments, the keys (distance << 22 | chunk-relative
// row) and the partial results of k_knn2, so the two kernels are interchangeable.  The waves do not cooperate:
// lane (h, r) of a wave loads 16 bytes of row r of each 32-row tile (the wave reads the tile's 1 KB exactly once)
// and expands them with 64 full-rate shift / mask instructions, no LDS, no barrier.  The 32 x 32 shape is chosen
// over 16 x 16 x 64 because of what issues beside it (tools/mfma_valu_probe.hip): six independent vector
// instructions per matrix instruction cost nothing next to the 32-cycle shape, while the 16-cycle shape hides
// two and then pays three cycles for every further one.
//
// A lane holds the results of 16 of a tile's 32 rows (ascending) for query lane % 32, so it keeps the two
// smallest keys over half of the rows; the two lanes of a query are merged at the end.  A key can only enter the
// lane's two smallest if its distance is below the lane's second-best distance: one wave vote per tile on the
// minima of the lanes' results skips the keyed folds (which are exact whatever the vote lets through).
#pragma once
#include <type_traits>

#include "knn2_kernel.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

#define KMM_THR_NONE 0x400  // above every d - popcount(q)
#define KMM_BIAS 256        // keys carry d - popcount(q) + 256 until the end
#define KMM_TILE 32         // train rows per tile

// four operand dwords of one instruction from the lane's word of the row: bits k0 .. k0 + 3 (+ 8 j) of it
__device__ __forceinline__ i32x4 kmm_bits01(uint32_t w, int k0) {
    return i32x4{(int)((w >> k0) & 0x01010101u), (int)((w >> (k0 + 1)) & 0x01010101u), (int)((w >> (k0 + 2)) & 0x01010101u),
                 (int)((w >> (k0 + 3)) & 0x01010101u)};
}

__global__ __launch_bounds__(256) void k_knn2_mm(const uint32_t* __restrict__ q, int nq, long long q_pair_words,
                                                 const uint32_t* __restrict__ t, int nt, long long t_pair_words,
                                                 int chunk_rows, int nchunks, int idx_base, uint2* __restrict__ part,
                                                 int32_t* __restrict__ o_idx, int32_t* __restrict__ o_d1,
                                                 int32_t* __restrict__ o_d2) {
    const int pair = blockIdx.z, chunk = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, col = lane & 31;
#ifdef KMM_EXP_CLOCK  // timing experiment: shader clock against the 100 MHz wall clock over one workgroup's life
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
#endif

    // ---- the lane's share of the two query tiles, as +1 / -1 bytes: instruction 2 i + u takes bits 4 u .. 4 u + 3
    // (+ 8 j) of word 4 h + i ----
    i32x4 bq[2][8];
#pragma unroll
    for (int qt = 0; qt < 2; qt++) {
        const int qi = blockIdx.x * 256 + wave * 64 + qt * 32 + col;
        const uint4 w4 = *(const uint4*)(q + (long long)pair * q_pair_words + (long long)min(qi, nq - 1) * 8 + 4 * h);
        const uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            const i32x4 b = kmm_bits01(w[ks >> 1], 4 * (ks & 1));
            int e[4];
#pragma unroll
            for (int i = 0; i < 4; i++) e[i] = (int)(((uint32_t)b[i] * 0xfeu) | 0x01010101u);  // bit ? -1 : +1
            bq[qt][ks] = i32x4{e[0], e[1], e[2], e[3]};
        }
    }

    const int j0 = chunk * chunk_rows;
    const int rows = min(nt - j0, chunk_rows);
    // the lane's 16 bytes of row r of the chunk: tp[r * 8 .. + 3] (rows are only 4-byte aligned)
    const uint32_t* __restrict__ tp = t + (long long)pair * t_pair_words + (long long)j0 * 8 + 4 * h;
    uint32_t best[2], best2[2];
    int thr[2];
#pragma unroll
    for (int qt = 0; qt < 2; qt++) {
        best[qt] = best2[qt] = KNN_NONE;
        thr[qt] = KMM_THR_NONE;
    }

    struct Row4 {
        uint32_t w[4];
    };
    auto load_row = [&](int r) {
        const uint32_t* p = tp + (long long)min(r, rows - 1) * 8;  // rows past the chunk re-read its last row
        Row4 v;
        v.w[0] = p[0];
        v.w[1] = p[1];
        v.w[2] = p[2];
        v.w[3] = p[3];
        return v;
    };
    // One tile of 32 rows starting at chunk row r0 whose bits are in `cur`.  TAIL = some of its rows lie past the
    // chunk (their keys are replaced by "none").
    auto tile = [&](const int r0, const Row4& cur, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        // accumulator input 0 (an inline constant, no registers to preset): acc[qt][reg] = d - popcount(q) for
        // row r0 + (reg & 3) + 8 (reg >> 2) + 4 h
        i32x16 acc[2];
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            const i32x4 a = kmm_bits01(cur.w[ks >> 1], 4 * (ks & 1));
#pragma unroll
            for (int qt = 0; qt < 2; qt++) {
                if (ks == 0)
                    acc[qt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[qt][0], i32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
                                                                    0, 0, 0);
                else
                    acc[qt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[qt][ks], acc[qt], 0, 0, 0);
            }
        }
        int below[2];
#pragma unroll
        for (int qt = 0; qt < 2; qt++) {
            int m = acc[qt][0];
#pragma unroll
            for (int i = 1; i < 16; i++) m = min(m, acc[qt][i]);
            below[qt] = m - thr[qt];  // negative: some result of the lane is below its second-best
        }
        const int rbase = r0 + 4 * h;
        // key = (acc + 256) << 22 | row: one shift-add per result
        const uint32_t kbase = ((uint32_t)KMM_BIAS << KNN_IDX_BITS) + (uint32_t)rbase;
#ifdef KMM_EXP_NOFOLD
        best[0] ^= (uint32_t)(below[0] | below[1]);  // timing experiment: no vote, no fold
        if (false) {
#else
        if (__builtin_amdgcn_ballot_w64((below[0] | below[1]) < 0) != 0) {
#endif
#pragma unroll
            for (int qt = 0; qt < 2; qt++) {
                if (__builtin_amdgcn_ballot_w64(below[qt] < 0) == 0) continue;
                // which of the sixteen results: all votes first (independent compares into scalar registers), then
                // a keyed fold for each result some lane needs.  The threshold only falls while the folds run, so
                // a vote taken before them lets through at least what it must.
                unsigned long long vote[16];
#pragma unroll
                for (int i = 0; i < 16; i++) vote[i] = __builtin_amdgcn_ballot_w64(acc[qt][i] < thr[qt]);
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    if (vote[i] == 0) continue;
                    const int ro = (i & 3) + 8 * (i >> 2);
                    uint32_t key = ((uint32_t)acc[qt][i] << KNN_IDX_BITS) + (kbase + (uint32_t)ro);
                    if (TAIL && rbase + ro >= rows) key = KNN_NONE;
                    fold_key(key, best[qt], best2[qt]);
                }
                thr[qt] = best2[qt] == KNN_NONE ? KMM_THR_NONE : (int)(best2[qt] >> KNN_IDX_BITS) - KMM_BIAS;
            }
        }
    };
    if (rows > 0) {
        // the bits of the next tile travel while the current one is multiplied
        const int nfull = rows / KMM_TILE;
        Row4 cur = load_row(col);
        for (int ti = 0; ti < nfull; ti++) {
            const Row4 nxt = load_row((ti + 1) * KMM_TILE + col);
            tile(ti * KMM_TILE, cur, std::false_type{});
            cur = nxt;
        }
        if (nfull * KMM_TILE < rows) tile(nfull * KMM_TILE, cur, std::true_type{});
    }

#ifdef KMM_EXP_CLOCK
    if (blockIdx.x == 0 && blockIdx.y == gridDim.y / 2 && tid == 0) {
        o_d2[nq - 1] = (int)(__builtin_readcyclecounter() - c0);
        o_d2[nq - 2] = (int)(wall_clock64() - w0);
    }
#endif
    // ---- the two lanes of a query (same lane % 32) merge their two smallest keys ----
#pragma unroll
    for (int qt = 0; qt < 2; qt++) {
        const uint32_t ob = (uint32_t)__shfl_xor((int)best[qt], 32), ob2 = (uint32_t)__shfl_xor((int)best2[qt], 32);
        fold_key(ob, best[qt], best2[qt]);
        fold_key(ob2, best[qt], best2[qt]);
        const int qi = blockIdx.x * 256 + wave * 64 + qt * 32 + col;
        if (h != 0 || qi >= nq) continue;
        // keys carry d - popcount(q) + 256: back to the distance
        const uint32_t* qp = q + (long long)pair * q_pair_words + (long long)qi * 8;
        int pcq = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) pcq += __popc(qp[w]);
        const uint32_t fix = (uint32_t)(pcq - KMM_BIAS) << KNN_IDX_BITS;
        const uint32_t b = best[qt] == KNN_NONE ? KNN_NONE : best[qt] + fix;
        const uint32_t b2 = best2[qt] == KNN_NONE ? KNN_NONE : best2[qt] + fix;
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
