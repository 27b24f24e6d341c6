// The Hamming k=2 scan on the matrix cores of gfx950.
//
// With the bits t_k of a train row as operand elements 0 / 1 and the bits q_k of a query as 1 - 2 q_k (+1 / -1), the
// dot product sum t_k (1 - 2 q_k) = popcount(t) - 2 popcount(t & q) is the Hamming distance minus popcount(q): an
// exact small integer that orders the rows of one query like the distance does.  The 4-bit float format of
// v_mfma_f32_32x32x64_f8f6f4 (E2M1: 0x2 = 1.0, 0xA = -1.0, 0 = 0) holds these values exactly and the fp32
// accumulators hold sums of up to 256 of them exactly, so four instructions give d - popcount(q) for 32 train rows
// x 32 queries: 8 distances per SIMD cycle at the instruction's issue rate, against the ~1.2 per cycle the xor +
// popcount instruction stream can issue (knn2_kernel.h), and the result is the same integer.  The order in which
// the instruction walks its products is irrelevant as long as both operands use the same bit -> (lane, nibble)
// rule; the rule here is the one that is cheapest to produce: the lanes of half h = lane / 32 own words 4 h .. 4 h + 3
// of a row, word i feeds instruction i, and nibble j of dword k of its four operand dwords is bit k + 4 j of the
// word: (word >> k) & 0x11111111, moved to bit 1 of the nibble.
//
// One workgroup = 256 queries (four waves x two 32-query tiles whose expanded bits stay in registers for the
// whole scan) against one chunk of train rows, with the arguments, the keys (distance << 22 | chunk-relative
// row) and the partial results of k_knn2, so the two kernels are interchangeable.  The waves do not cooperate:
// lane (h, r) of a wave loads 16 bytes of row r of each 32-row tile (the wave reads the tile's 1 KB exactly once)
// and expands them with 20 full-rate mask / shift instructions, no LDS, no barrier.  (Measured on the way here,
// tools/mfma_valu_probe.hip and tools/knn_mm_probe.hip: an LDS stage shared by the workgroup lost more at its
// barriers than the shared expansion saved; 8-bit integer operands need twice the instructions and twice the
// expansion; next to a 16 x 16 instruction only two independent vector instructions issue for free, next to a
// 32 x 32 one at least six; and the chip's clock falls to ~1.7 GHz under this load, so rates are quoted per second.)
//
// A lane holds the results of 16 of a tile's 32 rows (ascending) for query lane % 32, so it keeps the two
// smallest keys over half of the rows; the two lanes of a query are merged at the end.
//
// Keys are floats while a chunk is scanned: d - popcount(q) + row / 2^15, distinct and ordered like (distance, row)
// (sums of integers of magnitude <= 256 and a 15-bit fraction are exact in fp32: 24 significant bits, checked
// through the matrix pipe by tools/mfma_exact_probe.hip).  The two smallest keys of a lane are maintained by a
// branch-free two-slot sorted insertion: min, max, min per key.
//
// The first tiles of a chunk insert every result.  While a scan is young nearly every tile holds a result that
// enters some query's two smallest (a query's second-best improves about 2 / j times in tile j, and a wave carries
// 64 queries), so a test would be paid in every tile and skip nothing; there the row fraction rides in the
// accumulator input of the tile's first matrix instruction and a result IS the key (KMM_SORT_TILES_DEFAULT tiles,
// three full-rate instructions per result).
// After that a tile is tested first: a key can only enter the query's two smallest if it is below the query's
// second-best key so far (over both lanes of the query), so one wave vote on the minima of the lanes' results,
// then one vote per result, skip the insertions (which are exact whatever the votes let through).
//
// The row field of a float key holds 2^15 rows: longer chunks are scanned in epochs of 1024 tiles, at whose ends
// the float keys become the integer keys (distance << 22 | row) in which the result is reported, and the distance
// of the query's second-best integer key bounds the threshold of the next epoch.
#pragma once
#include <type_traits>

#include "knn2_kernel.h"

#define KMM_THR_NONE 0x400  // above every d - popcount(q)
#define KMM_BIAS 256        // keys carry d - popcount(q) + 256 until the end
#define KMM_TILE 32         // train rows per tile
#define KMM_FKEY_BITS 15    // row bits of the float keys (the fraction of the key)
#define KMM_FKEY_SCALE (1 << KMM_FKEY_BITS)
#define KMM_EPOCH_TILES (KMM_FKEY_SCALE / KMM_TILE)  // tiles whose rows fit the fraction
#define KMM_SORT_TILES_MAX KMM_EPOCH_TILES           // the always-insert phase stays inside the first epoch
#define KMM_SORT_TILES_DEFAULT 24

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// The four operand dwords of one instruction from the lane's word of a train row: nibble j of dword k stands for
// bit k + 4 j of the word.  The bit stays where it is inside its nibble wherever E2M1 has a value for that position
// (0x1 = 0.5, 0x2 = 1.0, 0x4 = 2.0; 0x8 is -0, so bit 3 moves to bit 2): four masks and one shift per word.  The
// query's nibble of the same position carries the reciprocal magnitude (kmm_query_nibbles), so every product is
// 0 or +-1 whatever the dword.
__device__ __forceinline__ i32x8 kmm_nibbles(uint32_t w) {
    return i32x8{(int)(w & 0x11111111u), (int)(w & 0x22222222u), (int)(w & 0x44444444u), (int)((w >> 1) & 0x44444444u),
                 0, 0, 0, 0};
}

// The query's side: +x for a clear bit, -x for a set one (nibble bit 3 is the sign), x = 2 / 1 / 0.5 / 0.5 for dwords
// 0 .. 3 (0x4, 0x2, 0x1, 0x1).
__device__ __forceinline__ i32x8 kmm_query_nibbles(uint32_t w) {
    return i32x8{(int)(((w << 3) & 0x88888888u) | 0x44444444u), (int)(((w << 2) & 0x88888888u) | 0x22222222u),
                 (int)(((w << 1) & 0x88888888u) | 0x11111111u), (int)((w & 0x88888888u) | 0x11111111u), 0, 0, 0, 0};
}

// The second smallest of the four keys the two lanes of a query (lane and lane ^ 32) hold as (best, best2), the same
// value in both lanes.  v_permlane32_swap exchanges the upper half of one register with the lower half of another;
// applied to two copies of a value it leaves "the lower lane's" in one and "the upper lane's" in the other, and the
// result is symmetric in the two, so no lane needs to know which half it is in (and nothing goes through LDS).
__device__ __forceinline__ uint32_t kmm_query_second(uint32_t best, uint32_t best2) {
    const auto b = __builtin_amdgcn_permlane32_swap(best, best, false, false);
    const auto b2 = __builtin_amdgcn_permlane32_swap(best2, best2, false, false);
    return min(max(b[0], b[1]), min(b2[0], b2[1]));
}
__device__ __forceinline__ float kmm_query_second_f(float k1, float k2) {
    const uint32_t u1 = __float_as_uint(k1), u2 = __float_as_uint(k2);
    const auto b = __builtin_amdgcn_permlane32_swap(u1, u1, false, false);
    const auto b2 = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
    return __builtin_fminf(__builtin_fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1])),
                           __builtin_fminf(__uint_as_float(b2[0]), __uint_as_float(b2[1])));
}

// A float key of an epoch that starts at (chunk-relative) row row0 as the integer key distance + KMM_BIAS << 22 | row.
__device__ __forceinline__ uint32_t kmm_integer_key(float fk, int row0) {
    if (fk == __builtin_inff()) return KNN_NONE;
    const float d = __builtin_floorf(fk);  // d - popcount(q); the fraction that is left is the row, exactly
    const uint32_t row = (uint32_t)((fk - d) * (float)KMM_FKEY_SCALE);
    return ((uint32_t)((int)d + KMM_BIAS) << KNN_IDX_BITS) + (uint32_t)row0 + row;
}

__global__ __launch_bounds__(256) void k_knn2_mm(const uint32_t* __restrict__ q, int nq, long long q_pair_words,
                                                  const uint32_t* __restrict__ t, int nt, long long t_pair_words,
                                                  int chunk_rows, int nchunks, int idx_base, uint2* __restrict__ part,
                                                  int32_t* __restrict__ o_idx, int32_t* __restrict__ o_d1,
                                                  int32_t* __restrict__ o_d2, int sort_tiles) {
    const int pair = blockIdx.z, chunk = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, col = lane & 31;
#ifdef KMM_EXP_CLOCK  // timing experiment: shader clock against the 100 MHz wall clock over one workgroup's life
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
#endif

    // ---- the lane's share of the two query tiles ----
    i32x8 bq[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; qt++) {
        const int qi = blockIdx.x * 256 + wave * 64 + qt * 32 + col;
        const uint4 w4 = *(const uint4*)(q + (long long)pair * q_pair_words + (long long)min(qi, nq - 1) * 8 + 4 * h);
        const uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int ks = 0; ks < 4; ks++) bq[qt][ks] = kmm_query_nibbles(w[ks]);
    }

    const int j0 = chunk * chunk_rows;
    const int rows = min(nt - j0, chunk_rows);
    const uint32_t* __restrict__ tp = t + (long long)pair * t_pair_words + (long long)j0 * 8 + 4 * h;
    uint32_t best[2], best2[2];  // integer keys of the epochs that are done
    float cap[2];                // what they allow the threshold of the current epoch to be
#pragma unroll
    for (int qt = 0; qt < 2; qt++) {
        best[qt] = best2[qt] = KNN_NONE;
        cap[qt] = (float)KMM_THR_NONE;
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
    // State of the current epoch: the lane's two smallest float keys and the query's threshold, in key units
    // relative to the epoch's first row.
    float fk1[2], fk2[2], thr[2];
    const float tile_step = (float)KMM_TILE / KMM_FKEY_SCALE;
    // Always-insert phase: ckey[i], the accumulator input of result register i, is the fraction of the register's row
    // counted from the epoch's first row (it advances by a tile per tile), so a result is the key.
    // Test phase: accumulator input 0, a result is the integer d - popcount(q); the fraction is added to the few
    // results that pass the votes: lane_frac (tile base + 4 h rows, one register) plus a constant per result register.
    f32x16 ckey;
    auto tile = [&](const int r0, const Row4& cur, const float tbase, auto tail_tag, auto sort_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        constexpr bool SORT = decltype(sort_tag)::value;
        // acc[qt][reg] = d - popcount(q) (+ ckey[reg]) for row r0 + (reg & 3) + 8 (reg >> 2) + 4 h
        f32x16 acc[2];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
#ifdef KMM_EXP_NOEXPAND  // timing experiment: the row's words as they are
            const i32x8 a = i32x8{(int)cur.w[ks], (int)cur.w[(ks + 1) & 3], (int)cur.w[(ks + 2) & 3], (int)cur.w[(ks + 3) & 3], 0, 0, 0, 0};
#else
            const i32x8 a = kmm_nibbles(cur.w[ks]);
#endif
#pragma unroll
            for (int qt = 0; qt < 2; qt++) {
                if (ks == 0 && SORT)
                    acc[qt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bq[qt][0], ckey, 4, 4, 0, 0, 0, 0);
                else if (ks == 0)
                    acc[qt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                        a, bq[qt][0], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 4, 4, 0, 0,
                        0, 0);
                else
                    acc[qt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bq[qt][ks], acc[qt], 4, 4, 0, 0, 0, 0);
            }
        }
        if constexpr (SORT) {
            // every result goes through the insertion: three full-rate instructions, no vote, no branch
#pragma unroll
            for (int i = 0; i < 16; i++) {
#pragma unroll
                for (int qt = 0; qt < 2; qt++) {
                    const float k = acc[qt][i];
                    fk2[qt] = __builtin_fminf(fk2[qt], __builtin_fmaxf(fk1[qt], k));
                    fk1[qt] = __builtin_fminf(fk1[qt], k);
                }
            }
            ckey += tile_step;
            return;
        }
        // A key is its result + lane_frac + the constant of its register; key < threshold implies
        // result < threshold - lane_frac, so the votes on that let through a superset (results whose distance ties
        // with the threshold's inside this very tile), and the insertion is exact whatever they let through.
        const float lane_frac = tbase + (float)(4 * h) * (1.f / KMM_FKEY_SCALE);
        float below[2], thr_rel[2];
#pragma unroll
        for (int qt = 0; qt < 2; qt++) {
            float m = acc[qt][0];
#pragma unroll
            for (int i = 1; i < 16; i++) m = fminf(m, acc[qt][i]);
            thr_rel[qt] = thr[qt] - lane_frac;  // exact: a multiple of 2^-15 below 2^9
            below[qt] = m - thr_rel[qt];        // negative: some key of the lane may be below the query's second-best
        }
#ifdef KMM_EXP_NOFOLD
        fk1[0] += fminf(below[0], below[1]);  // timing experiment: no vote, no insertion
        if (false) {
#else
        if (__builtin_amdgcn_ballot_w64(fminf(below[0], below[1]) < 0.f) != 0) {
#endif
            const int rbase = r0 + 4 * h;
#pragma unroll
            for (int qt = 0; qt < 2; qt++) {
                if (__builtin_amdgcn_ballot_w64(below[qt] < 0.f) == 0) continue;
                unsigned long long vote[16];
#pragma unroll
                for (int i = 0; i < 16; i++) vote[i] = __builtin_amdgcn_ballot_w64(acc[qt][i] < thr_rel[qt]);
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    if (vote[i] == 0) continue;
                    const int ro = (i & 3) + 8 * (i >> 2);
                    float k = (acc[qt][i] + lane_frac) + (float)ro * (1.f / KMM_FKEY_SCALE);
                    if (TAIL && rbase + ro >= rows) k = __builtin_inff();
                    fk2[qt] = __builtin_fminf(fk2[qt], __builtin_fmaxf(fk1[qt], k));
                    fk1[qt] = __builtin_fminf(fk1[qt], k);
                }
                // The threshold is the second-smallest key of the QUERY, i.e. over both lanes that share it
                thr[qt] = __builtin_fminf(cap[qt], kmm_query_second_f(fk1[qt], fk2[qt]));
            }
        }
    };
    if (rows > 0) {
        const int nfull = rows / KMM_TILE;
        const int ntiles = (rows + KMM_TILE - 1) / KMM_TILE;
        Row4 cur = load_row(col);
        int ti = 0;
#pragma unroll
        for (int qt = 0; qt < 2; qt++) fk1[qt] = fk2[qt] = __builtin_inff();
        {
            // the always-insert phase, inside the first epoch
#pragma unroll
            for (int i = 0; i < 16; i++) ckey[i] = (float)(4 * h + (i & 3) + 8 * (i >> 2)) * (1.f / KMM_FKEY_SCALE);
            const int nsort = min(nfull, sort_tiles);
            for (; ti < nsort; ti++) {
                const Row4 nxt = load_row((ti + 1) * KMM_TILE + col);
                tile(ti * KMM_TILE, cur, 0.f, std::false_type{}, std::true_type{});
                cur = nxt;
            }
        }
        for (int e0 = 0; e0 < ntiles; e0 += KMM_EPOCH_TILES) {
            const int e1 = min(ntiles, e0 + KMM_EPOCH_TILES), efull = min(nfull, e1);
#pragma unroll
            for (int qt = 0; qt < 2; qt++) thr[qt] = __builtin_fminf(cap[qt], kmm_query_second_f(fk1[qt], fk2[qt]));
            float tbase = (float)(ti - e0) * tile_step;  // fraction of the tile's first row
            for (; ti < efull; ti++) {
                const Row4 nxt = load_row((ti + 1) * KMM_TILE + col);
                tile(ti * KMM_TILE, cur, tbase, std::false_type{}, std::false_type{});
                cur = nxt;
                tbase += tile_step;
            }
            if (efull < e1) tile(efull * KMM_TILE, cur, tbase, std::true_type{}, std::false_type{});
            // the epoch's float keys join the integer keys; the next epoch's keys count rows from its own start, so
            // all it keeps of this one is a bound: the distance of the query's second-best key (a later row of the
            // same distance loses to it)
#pragma unroll
            for (int qt = 0; qt < 2; qt++) {
                fold_key(kmm_integer_key(fk1[qt], e0 * KMM_TILE), best[qt], best2[qt]);
                fold_key(kmm_integer_key(fk2[qt], e0 * KMM_TILE), best[qt], best2[qt]);
                if (e1 < ntiles) {
                    const uint32_t u2 = kmm_query_second(best[qt], best2[qt]);
                    cap[qt] = u2 == KNN_NONE ? (float)KMM_THR_NONE : (float)((int)(u2 >> KNN_IDX_BITS) - KMM_BIAS);
                    fk1[qt] = fk2[qt] = __builtin_inff();
                }
            }
        }
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
