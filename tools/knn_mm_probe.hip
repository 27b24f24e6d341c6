// Times variants of the matrix-core scan kernel (orb-slam2_amd/csrc/knn2_mfma.h) on one map-shard-shaped problem
// without the library around it (-DKMM_EXP_NOFOLD: no vote, no fold; -DKMM_EXP_CLOCK: the shader clock the chip
// holds under the kernel), and with -DKMM_RATE the bare issue rate of v_mfma_i32_16x16x64_i8 chains.
// Results of the variants are not meaningful; only the time is.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
namespace {
#include "../orb-slam2_amd/csrc/knn2_mfma.h"
}
typedef int i32x4 __attribute__((ext_vector_type(4)));
// NACC independent accumulator chains per wave, WAVES_PER_SIMD waves: cycles per instruction on every SIMD
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_rate(int iters, int* out) {
    i32x4 a = i32x4{(int)threadIdx.x, 1, 2, 3}, b = i32x4{4, 5, (int)blockIdx.x, 7};
    i32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = i32x4{i, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 0x7fffffff) out[0] = s;
}
template <int NACC>
static void rate(const hipDeviceProp_t& prop, int blocks_per_cu, int* o) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma_rate<NACC>, dim3(prop.multiProcessorCount * blocks_per_cu), dim3(256), 0, 0, iters, o);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // per SIMD: blocks_per_cu waves, each iters * NACC instructions
    const double clk = prop.clockRate * 1e3;
    printf("mfma_i32_16x16x64_i8: %d chains x %d waves per SIMD: %.1f cycles per instruction (clock %.0f MHz nominal)\n", NACC,
           blocks_per_cu, best * 1e-3 * clk / ((double)iters * NACC * blocks_per_cu), clk / 1e6);
}
int main(int argc, char** argv) {
    const int nq = 4096, nt = argc > 1 ? atoi(argv[1]) : 1250000, bpc = argc > 2 ? atoi(argv[2]) : 16;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int qblocks = nq / 256;
    int nchunks = bpc * prop.multiProcessorCount / qblocks;
    if (nchunks < 1) nchunks = 1;
    const int chunk_rows = ((nt + nchunks - 1) / nchunks + 31) & ~31;
    nchunks = (nt + chunk_rows - 1) / chunk_rows;
    std::vector<uint32_t> hq((size_t)nq * 8), ht((size_t)nt * 8);
    uint32_t x = 12345;
    for (auto& v : hq) v = (x = x * 1664525u + 1013904223u) ^ (x >> 13);
    for (auto& v : ht) v = (x = x * 1664525u + 1013904223u) ^ (x >> 11);
    uint32_t *q, *t;
    uint2* part;
    int32_t* o;
    hipMalloc(&q, hq.size() * 4);
    hipMalloc(&t, ht.size() * 4);
    hipMalloc(&part, (size_t)nchunks * nq * 8);
    hipMalloc(&o, (size_t)nq * 12);
    hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(t, ht.data(), ht.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 8; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_knn2_mm, dim3(qblocks, nchunks, 1), dim3(256), 0, 0, q, nq, 0LL, t, nt, 0LL, chunk_rows, nchunks, 0, part,
                           o, o + nq, o + 2 * nq, KMM_SORT_TILES_DEFAULT);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double clk = prop.clockRate * 1e3;
    printf("%s nt=%d chunks=%d x %d rows: %.3f ms  %.3e comparisons/s  %.1f nominal SIMD-cycles per 64 distances\n", VARIANT, nt,
           nchunks, chunk_rows, best, (double)nq * nt / (best * 1e-3),
           best * 1e-3 * clk * prop.multiProcessorCount * 4 / ((double)nq * nt / 64));
#ifdef KMM_EXP_CLOCK
    {
        int h2[2];
        (void)hipMemcpy(h2, o + 2 * nq + nq - 2, 8, hipMemcpyDeviceToHost);
        printf("one workgroup: %d shader cycles in %d ticks of the 100 MHz clock = %.0f MHz\n", h2[1], h2[0], h2[1] / (h2[0] / 100.0));
    }
#endif
#ifdef KMM_RATE
    rate<1>(prop, 1, o);
    rate<4>(prop, 1, o);
    rate<1>(prop, 4, o);
    rate<4>(prop, 4, o);
#endif
    return 0;
}
