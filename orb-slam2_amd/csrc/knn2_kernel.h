// The Hamming k=2 scan kernel of the matcher (see matcher.hip for the mapping and the launch plan).
// Kept in a header so that tools/knn_kernel_probe.hip can time variants of the same source.
#pragma once
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>

#define KNN_IDX_BITS 22
#define KNN_IDX_MASK ((1u << KNN_IDX_BITS) - 1u)
#define KNN_NONE 0xffffffffu

// popcount(x) + acc in one instruction; the explicit form keeps the 8-word distance a chain of
// 8 v_xor_b32 + 8 v_bcnt_u32_b32 (the compiler otherwise splits the sum into a tree with extra adds).
// The s_nop after it is a measured issue-order fact of gfx950 (tools/nop_probe.hip): v_bcnt_u32_b32 occupies
// the SIMD for 4 cycles, v_xor_b32 for 2, but in a stream that mixes the two every instruction costs 4 unless
// each 4-cycle instruction is followed by an s_nop 0 (another wave then issues in the freed slot):
// 8.5 -> 6.3 cycles per xor + popcount pair.
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2\n\ts_nop 0" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ uint32_t ham8(const uint32_t q[8], const uint32_t* __restrict__ t) {
    uint32_t d = bcnt_acc(q[0] ^ t[0], 0u);
    d = bcnt_acc(q[1] ^ t[1], d);
    d = bcnt_acc(q[2] ^ t[2], d);
    d = bcnt_acc(q[3] ^ t[3], d);
    d = bcnt_acc(q[4] ^ t[4], d);
    d = bcnt_acc(q[5] ^ t[5], d);
    d = bcnt_acc(q[6] ^ t[6], d);
    d = bcnt_acc(q[7] ^ t[7], d);
    return d;
}

// Folds one candidate key into the running two smallest keys (best <= best2 always holds, so the
// new second smallest is the median of the three).
__device__ __forceinline__ void fold_key(uint32_t key, uint32_t& best, uint32_t& best2) {
    best2 = med3_u32(best, key, best2);
    best = min(best, key);
}

#define KNN_GROUP 8                 // rows per filter decision
#define KNN_HALF 4                  // rows per scalar-load batch (two 64-byte scalar loads)
#define KNN_WARM_ROWS 512           // multiple of KNN_GROUP: rows of a chunk folded unconditionally
#define KNN_BLOCKS_PER_CU 16        // workgroups per CU the launch plan aims at: two rounds of 8 resident ones (measured best)
#define KNN_THR_NONE 0x200u         // above every distance: no second-best yet

// Four wave-uniform train rows (32 words) in scalar registers, fetched as two 64-byte scalar loads.
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
struct RowQuad {
    u32x16 lo, hi;  // rows 0-1, rows 2-3
};
__device__ __forceinline__ void load_quad(RowQuad& g, const uint32_t* __restrict__ p) {
    g.lo = *(const u32x16*)p;
    g.hi = *(const u32x16*)(p + 16);
}

// One word of four rows: the issue order measured to be the fast one on gfx950 (tools/nop_probe.hip).
// v_bcnt_u32_b32 occupies the SIMD for 4 cycles, v_xor_b32 for 2, but in a stream that mixes them every
// instruction costs 4 unless each popcount is followed by an s_nop 0 (another wave issues in the freed slot),
// and xors that read a scalar register must not follow one another directly: xor, popcount, nop, four
// independent chains -> 6.3 cycles per pair instead of 8.5.
#define KNN_WORD_STEP(QW, S0, S1, S2, S3)                                                               \
    asm volatile(                                                                                       \
        "v_xor_b32 %4, %8, %12\n\tv_bcnt_u32_b32 %0, %4, %0\n\ts_nop 0\n\t"                                \
        "v_xor_b32 %5, %9, %12\n\tv_bcnt_u32_b32 %1, %5, %1\n\ts_nop 0\n\t"                                \
        "v_xor_b32 %6, %10, %12\n\tv_bcnt_u32_b32 %2, %6, %2\n\ts_nop 0\n\t"                               \
        "v_xor_b32 %7, %11, %12\n\tv_bcnt_u32_b32 %3, %7, %3\n\ts_nop 0"                                   \
        : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)    \
        : "s"(S0), "s"(S1), "s"(S2), "s"(S3), "v"(QW))

// Distances of four rows to the lane's query, each starting at `bias` (0, or minus the lane's threshold:
// the sign bit of the result then says "distance < threshold" and no instruction is spent on the comparison).
__device__ __forceinline__ void ham8_quad(const uint32_t q[8], const RowQuad& g, uint32_t bias, uint32_t* d) {
    uint32_t x0, x1, x2, x3;
#pragma unroll
    for (int u = 0; u < KNN_HALF; u++) d[u] = bias;
    KNN_WORD_STEP(q[0], g.lo[0], g.lo[8], g.hi[0], g.hi[8]);
    KNN_WORD_STEP(q[1], g.lo[1], g.lo[9], g.hi[1], g.hi[9]);
    KNN_WORD_STEP(q[2], g.lo[2], g.lo[10], g.hi[2], g.hi[10]);
    KNN_WORD_STEP(q[3], g.lo[3], g.lo[11], g.hi[3], g.hi[11]);
    KNN_WORD_STEP(q[4], g.lo[4], g.lo[12], g.hi[4], g.hi[12]);
    KNN_WORD_STEP(q[5], g.lo[5], g.lo[13], g.hi[5], g.hi[13]);
    KNN_WORD_STEP(q[6], g.lo[6], g.lo[14], g.hi[6], g.hi[14]);
    KNN_WORD_STEP(q[7], g.lo[7], g.lo[15], g.hi[7], g.hi[15]);
}

// What a group of eight rows does to the lane's two smallest keys, given the rows' (biased) distances e[].
// WARM (bias 0): every row is folded.  Otherwise (steady state, bias -thr with thr = the lane's second-best
// distance): a row almost never enters a lane's top two any more; the sign bit of e[u] says "distance < thr",
// the eight results are OR-ed and ONE sign test per group decides, for the whole wave, whether any lane needs
// the exact keyed fold.  A key (distance << 22 | row) can only change (best, best2) if it is smaller than
// best2; rows ascend, so that is exactly distance < thr.  The slow path folds with the true keys, so results
// do not depend on the filter; a threshold that is stale inside a group only lets more rows through.
template <bool WARM>
__device__ __forceinline__ void finish_group(const uint32_t e[KNN_GROUP], int row0, uint32_t& best, uint32_t& best2,
                                             uint32_t& thr) {
    if (WARM) {
#pragma unroll
        for (int u = 0; u < KNN_GROUP; u++) fold_key((e[u] << KNN_IDX_BITS) | (uint32_t)(row0 + u), best, best2);
        thr = best2 >> KNN_IDX_BITS;
        return;
    }
    uint32_t any = e[0];
#pragma unroll
    for (int u = 1; u < KNN_GROUP; u++) any |= e[u];
#ifdef KNN_EXP_NOBRANCH
    best ^= any;  // experiment: no filter decision at all
    if (false) {
#else
    if (__builtin_amdgcn_ballot_w64((int)any < 0) != 0) {
#endif
        // all eight wave votes first (independent compares into scalar registers), then scalar tests
        unsigned long long vote[KNN_GROUP];
#pragma unroll
        for (int u = 0; u < KNN_GROUP; u++) vote[u] = __builtin_amdgcn_ballot_w64((int)e[u] < 0);
#pragma unroll
        for (int u = 0; u < KNN_GROUP; u++) {
            if (vote[u] != 0) fold_key(((e[u] + thr) << KNN_IDX_BITS) | (uint32_t)(row0 + u), best, best2);
        }
        thr = best2 >> KNN_IDX_BITS;
    }
}

// Scalar loads return out of order, so every wait for one is a wait for all.  Naming a register of a quad
// here makes the compiler place that wait BEFORE the next quad's loads are issued (it would otherwise sit
// right behind them and expose their latency): wait, issue the next loads, compare the current quad.
__device__ __forceinline__ void quad_arrived(const RowQuad& g) { asm volatile("" ::"s"(g.lo[0]), "s"(g.hi[15])); }

// grid: (query blocks of 256, train chunks, pairs).  Each block scans rows [chunk*chunk_rows,
// min(nt, (chunk+1)*chunk_rows)) and emits its two smallest keys per query.  With one chunk the
// final (idx, d1, d2) are written directly.  The row quads are double-buffered in scalar registers:
// the loads of the next group are in flight while the current one is compared.
// (KNN_HELPERS_ONLY: a translation unit that wants the keys and folds above but not this kernel)
#ifndef KNN_HELPERS_ONLY
#ifndef KNN_KERNEL_NAME
#define KNN_KERNEL_NAME k_knn2
#endif
__global__ __launch_bounds__(256) void KNN_KERNEL_NAME(const uint32_t* __restrict__ q, int nq, long long q_pair_words,
                                              const uint32_t* __restrict__ t, int nt, long long t_pair_words,
                                              int chunk_rows, int nchunks, int idx_base, uint2* __restrict__ part,
                                              int32_t* __restrict__ o_idx, int32_t* __restrict__ o_d1,
                                              int32_t* __restrict__ o_d2) {
    const int pair = blockIdx.z, chunk = blockIdx.y;
    const int qi = blockIdx.x * 256 + threadIdx.x;
    uint32_t qw[8];
    {
        const uint32_t* qp = q + (long long)pair * q_pair_words + (long long)min(qi, nq - 1) * 8;
        const uint4 a = *(const uint4*)qp, b = *(const uint4*)(qp + 4);
        qw[0] = a.x; qw[1] = a.y; qw[2] = a.z; qw[3] = a.w;
        qw[4] = b.x; qw[5] = b.y; qw[6] = b.z; qw[7] = b.w;
    }
    const int j0 = chunk * chunk_rows;
    const int rows = min(nt - j0, chunk_rows);
    const uint32_t* __restrict__ tp = t + (long long)pair * t_pair_words + (long long)j0 * 8;
    uint32_t best = KNN_NONE, best2 = KNN_NONE, thr = KNN_THR_NONE;
    const int full = rows & ~(KNN_GROUP - 1);  // rows covered by whole groups
    int j = 0;
    if (full > 0) {
        // two quads ping-pong in scalar registers: while one is compared the other one's loads are in flight
        RowQuad ga, gb;
        load_quad(ga, tp);
        for (; j < full; j += KNN_GROUP) {
            const bool warm = j < KNN_WARM_ROWS;
            const uint32_t bias = warm ? 0u : 0u - thr;
            uint32_t e[KNN_GROUP];
            quad_arrived(ga);
            load_quad(gb, tp + (long long)(j + KNN_HALF) * 8);
            ham8_quad(qw, ga, bias, e);
            quad_arrived(gb);
            // the quad after the last group is a re-read of the last quad instead of a read past the chunk
            load_quad(ga, tp + (long long)min(j + KNN_GROUP, full - KNN_HALF) * 8);
            ham8_quad(qw, gb, bias, e + KNN_HALF);
            if (warm) finish_group<true>(e, j, best, best2, thr);
            else finish_group<false>(e, j, best, best2, thr);
        }
    }
    for (; j < rows; j++) fold_key((ham8(qw, tp + (long long)j * 8) << KNN_IDX_BITS) | (uint32_t)j, best, best2);
    if (qi >= nq) return;
    if (nchunks == 1) {
        const long long o = (long long)pair * nq + qi;
        o_idx[o] = best == KNN_NONE ? -1 : (int)(best & KNN_IDX_MASK) + idx_base;
        o_d1[o] = best == KNN_NONE ? INT_MAX : (int)(best >> KNN_IDX_BITS);
        o_d2[o] = best2 == KNN_NONE ? INT_MAX : (int)(best2 >> KNN_IDX_BITS);
    } else {
        part[((long long)pair * nchunks + chunk) * nq + qi] = make_uint2(best, best2);
    }
}
#endif  // KNN_HELPERS_ONLY
