// Issue-order probe for the Hamming kNN inner loop: how many SIMD cycles does one wave64 "row"
// (8 v_xor_b32 with an SGPR operand + 8 accumulating v_bcnt_u32_b32) cost, depending on how the two
// instruction classes are ordered?  No memory traffic: the row words are kernel arguments.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define ITERS 512

// one row, all xors first, then the dependent bcnt chain
#define ROW_GROUPED(d, acc)                                                                                   \
    asm volatile(                                                                                             \
        "v_xor_b32 %1, %17, %9\n v_xor_b32 %2, %18, %10\n v_xor_b32 %3, %19, %11\n v_xor_b32 %4, %20, %12\n"  \
        "v_xor_b32 %5, %21, %13\n v_xor_b32 %6, %22, %14\n v_xor_b32 %7, %23, %15\n v_xor_b32 %8, %24, %16\n" \
        "v_bcnt_u32_b32 %0, %1, %25\n v_bcnt_u32_b32 %0, %2, %0\n v_bcnt_u32_b32 %0, %3, %0\n"               \
        "v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %0, %5, %0\n v_bcnt_u32_b32 %0, %6, %0\n"                \
        "v_bcnt_u32_b32 %0, %7, %0\n v_bcnt_u32_b32 %0, %8, %0\n"                                            \
        : "=&v"(d), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7)    \
        : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7), "s"(t0), "s"(t1), "s"(t2), \
          "s"(t3), "s"(t4), "s"(t5), "s"(t6), "s"(t7), "v"(acc))
#define ROW_GROUPED_V(d, acc)                                                                                   \
    asm volatile(                                                                                             \
        "v_xor_b32 %1, %17, %9\n v_xor_b32 %2, %18, %10\n v_xor_b32 %3, %19, %11\n v_xor_b32 %4, %20, %12\n"  \
        "v_xor_b32 %5, %21, %13\n v_xor_b32 %6, %22, %14\n v_xor_b32 %7, %23, %15\n v_xor_b32 %8, %24, %16\n" \
        "v_bcnt_u32_b32 %0, %1, %25\n v_bcnt_u32_b32 %0, %2, %0\n v_bcnt_u32_b32 %0, %3, %0\n"               \
        "v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %0, %5, %0\n v_bcnt_u32_b32 %0, %6, %0\n"                \
        "v_bcnt_u32_b32 %0, %7, %0\n v_bcnt_u32_b32 %0, %8, %0\n"                                            \
        : "=&v"(d), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7)    \
        : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7), "v"(v0), "v"(v1), "v"(v2), \
          "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7), "v"(acc))
// strictly alternating xor / bcnt
#define ROW_ALTERNATING(d, acc)                                                                               \
    asm volatile(                                                                                             \
        "v_xor_b32 %1, %17, %9\n v_bcnt_u32_b32 %0, %1, %25\n v_xor_b32 %2, %18, %10\n v_bcnt_u32_b32 %0, %2, %0\n" \
        "v_xor_b32 %3, %19, %11\n v_bcnt_u32_b32 %0, %3, %0\n v_xor_b32 %4, %20, %12\n v_bcnt_u32_b32 %0, %4, %0\n" \
        "v_xor_b32 %5, %21, %13\n v_bcnt_u32_b32 %0, %5, %0\n v_xor_b32 %6, %22, %14\n v_bcnt_u32_b32 %0, %6, %0\n" \
        "v_xor_b32 %7, %23, %15\n v_bcnt_u32_b32 %0, %7, %0\n v_xor_b32 %8, %24, %16\n v_bcnt_u32_b32 %0, %8, %0\n" \
        : "=&v"(d), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7)    \
        : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7), "s"(t0), "s"(t1), "s"(t2), \
          "s"(t3), "s"(t4), "s"(t5), "s"(t6), "s"(t7), "v"(acc))
// xors in pairs: xor xor bcnt bcnt
#define ROW_PAIRS(d, acc)                                                                                     \
    asm volatile(                                                                                             \
        "v_xor_b32 %1, %17, %9\n v_xor_b32 %2, %18, %10\n v_bcnt_u32_b32 %0, %1, %25\n v_bcnt_u32_b32 %0, %2, %0\n" \
        "v_xor_b32 %3, %19, %11\n v_xor_b32 %4, %20, %12\n v_bcnt_u32_b32 %0, %3, %0\n v_bcnt_u32_b32 %0, %4, %0\n" \
        "v_xor_b32 %5, %21, %13\n v_xor_b32 %6, %22, %14\n v_bcnt_u32_b32 %0, %5, %0\n v_bcnt_u32_b32 %0, %6, %0\n" \
        "v_xor_b32 %7, %23, %15\n v_xor_b32 %8, %24, %16\n v_bcnt_u32_b32 %0, %7, %0\n v_bcnt_u32_b32 %0, %8, %0\n" \
        : "=&v"(d), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7)    \
        : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7), "s"(t0), "s"(t1), "s"(t2), \
          "s"(t3), "s"(t4), "s"(t5), "s"(t6), "s"(t7), "v"(acc))
// two independent half chains (4 + 4 words), summed at the end with a full-rate add
#define ROW_TWOCHAIN(d, acc)                                                                                  \
    asm volatile(                                                                                             \
        "v_xor_b32 %1, %17, %9\n v_xor_b32 %2, %18, %10\n v_xor_b32 %3, %19, %11\n v_xor_b32 %4, %20, %12\n"  \
        "v_xor_b32 %5, %21, %13\n v_xor_b32 %6, %22, %14\n v_xor_b32 %7, %23, %15\n v_xor_b32 %8, %24, %16\n" \
        "v_bcnt_u32_b32 %0, %1, %25\n v_bcnt_u32_b32 %1, %2, %3\n v_bcnt_u32_b32 %0, %4, %0\n"               \
        "v_bcnt_u32_b32 %1, %5, %1\n v_bcnt_u32_b32 %0, %6, %0\n v_bcnt_u32_b32 %1, %7, %1\n"                \
        "v_bcnt_u32_b32 %0, %8, %0\n v_add_u32 %0, %0, %1\n"                                                 \
        : "=&v"(d), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7)    \
        : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7), "s"(t0), "s"(t1), "s"(t2), \
          "s"(t3), "s"(t4), "s"(t5), "s"(t6), "s"(t7), "v"(acc))

#define KERNEL(name, ROW, EXTRA)                                                                              \
    __global__ __launch_bounds__(256) void name(uint32_t* out, uint32_t t0, uint32_t t1, uint32_t t2,         \
                                                uint32_t t3, uint32_t t4, uint32_t t5, uint32_t t6, uint32_t t7) { \
        uint32_t q0 = threadIdx.x, q1 = q0 * 3, q2 = q0 * 5, q3 = q0 * 7, q4 = q0 * 11, q5 = q0 * 13,        \
                 q6 = q0 * 17, q7 = q0 * 19;                                                                  \
        uint32_t x0, x1, x2, x3, x4, x5, x6, x7, d, any = 0, acc = blockIdx.x;                                \
        for (int i = 0; i < ITERS; i++) {                                                                     \
            ROW(d, acc);                                                                                      \
            EXTRA;                                                                                            \
        }                                                                                                     \
        if ((any ^ d) == 0x12345u) out[0] = d;                                                                \
    }

KERNEL(k_grouped, ROW_GROUPED, acc = d)
KERNEL(k_alternating, ROW_ALTERNATING, acc = d)
KERNEL(k_pairs, ROW_PAIRS, acc = d)
KERNEL(k_twochain, ROW_TWOCHAIN, acc = d)
KERNEL(k_grouped_or, ROW_GROUPED, asm volatile("v_or_b32 %0, %0, %1" : "+v"(any) : "v"(d)))
KERNEL(k_grouped_fold, ROW_GROUPED,
       asm volatile("v_lshl_or_b32 %1, %1, 22, %2\n v_med3_u32 %0, %3, %1, %0\n v_min_u32 %3, %3, %1"
                    : "+v"(any), "+v"(d), "+s"(t0), "+v"(acc)))

// rows replicated in VGPRs (set up once): what the xor costs without the scalar operand
__global__ __launch_bounds__(256) void k_grouped_vgpr(uint32_t* out, uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3,
                                                      uint32_t t4, uint32_t t5, uint32_t t6, uint32_t t7) {
    uint32_t q0 = threadIdx.x, q1 = q0 * 3, q2 = q0 * 5, q3 = q0 * 7, q4 = q0 * 11, q5 = q0 * 13, q6 = q0 * 17, q7 = q0 * 19;
    uint32_t v0, v1, v2, v3, v4, v5, v6, v7;
    asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %9\n v_mov_b32 %2, %10\n v_mov_b32 %3, %11\n v_mov_b32 %4, %12\n"
                 "v_mov_b32 %5, %13\n v_mov_b32 %6, %14\n v_mov_b32 %7, %15"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7)
                 : "s"(t0), "s"(t1), "s"(t2), "s"(t3), "s"(t4), "s"(t5), "s"(t6), "s"(t7));
    uint32_t x0, x1, x2, x3, x4, x5, x6, x7, d, acc = blockIdx.x;
    for (int i = 0; i < ITERS; i++) {
        ROW_GROUPED_V(d, acc);
        acc = d;
    }
    if (d == 0x12345u) out[0] = d;
}

// rows broadcast from LDS into VGPRs (two ds_read_b128 per row), NQ queries per lane share them
template <int NQ, int PAD = 0>
__global__ __launch_bounds__(256) void k_lds_rows(uint32_t* out, uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3,
                                                  uint32_t t4, uint32_t t5, uint32_t t6, uint32_t t7) {
    __shared__ uint4 rows[512 + PAD];  // PAD: extra LDS that only limits how many workgroups fit a CU
    for (int i = threadIdx.x; i < 512 + PAD; i += 256) rows[i] = make_uint4(t0 + i, t1 ^ i, t2 * i, t3 + 3 * i);
    __syncthreads();
    uint32_t q[NQ][8];
    for (int n = 0; n < NQ; n++)
        for (int k = 0; k < 8; k++) q[n][k] = threadIdx.x * (2 * k + 3) + n * 977 + t4;
    uint32_t acc[NQ];
    for (int n = 0; n < NQ; n++) acc[n] = n;
    for (int i = 0; i < ITERS / NQ; i++) {
        const uint4 a = rows[(2 * i) & 511], b = rows[(2 * i + 1) & 511];
        const uint32_t v0 = a.x, v1 = a.y, v2 = a.z, v3 = a.w, v4 = b.x, v5 = b.y, v6 = b.z, v7 = b.w;
#pragma unroll
        for (int n = 0; n < NQ; n++) {
            const uint32_t q0 = q[n][0], q1 = q[n][1], q2 = q[n][2], q3 = q[n][3], q4 = q[n][4], q5 = q[n][5], q6 = q[n][6],
                           q7 = q[n][7];
            uint32_t x0, x1, x2, x3, x4, x5, x6, x7, d;
            ROW_GROUPED_V(d, acc[n]);
            acc[n] = d;
        }
    }
    uint32_t s = 0;
    for (int n = 0; n < NQ; n++) s ^= acc[n];
    if (s == 0x12345u) out[0] = s;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double clk = prop.clockRate * 1e3;
    const int cus = prop.multiProcessorCount;
    uint32_t* out;
    hipMalloc(&out, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    typedef void (*K)(uint32_t*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t);
    auto run = [&](const char* name, K k, int waves_per_simd) {
        const int blocks = cus * waves_per_simd * 8;  // 4 waves per block, one per SIMD
        float best = 1e9f;
        for (int r = 0; r < 4; r++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1u + r, 2u, 3u, 4u, 5u, 6u, 7u, 8u);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double rows = (double)blocks * 4 * ITERS;  // wave-rows
        printf("%-16s %d waves/SIMD resident x8 rounds: %7.3f ms  %.2f SIMD-cycles per wave-row\n", name, waves_per_simd, best,
               best * 1e-3 * clk * cus * 4.0 / rows);
    };
    for (int w : {8}) {
        run("grouped", k_grouped, w);
        run("alternating", k_alternating, w);
        run("pairs", k_pairs, w);
        run("twochain", k_twochain, w);
        run("grouped+or", k_grouped_or, w);
        run("grouped+fold", k_grouped_fold, w);
        run("grouped vgpr", k_grouped_vgpr, w);
        run("lds rows NQ=1", k_lds_rows<1>, w);
        run("lds NQ=1 4w/SIMD", k_lds_rows<1, 2048>, w);
        run("lds NQ=1 2w/SIMD", k_lds_rows<1, 4096>, w);
        run("lds NQ=1 1w/SIMD", k_lds_rows<1, 9000>, w);
        run("lds rows NQ=2", k_lds_rows<2>, w);
        run("lds rows NQ=4", k_lds_rows<4>, w);
    }
    return 0;
}
