// Times variants of the matcher's scan kernel (orb-slam2_amd/csrc/knn2_kernel.h) on one map-shard-shaped
// problem without the library around it: -DKNN_EXP_NOBRANCH (no filter decision), -DKNN_EXP_NOLDS (rows stay in
// registers).  Results of the variants are not meaningful; only the time is.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
namespace {
#include "../orb-slam2_amd/csrc/knn2_kernel.h"
}
int main(int argc, char** argv) {
    const int nq = 4096, nt = argc > 1 ? atoi(argv[1]) : 1250000, bpc = argc > 2 ? atoi(argv[2]) : 4;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int qblocks = nq / 256;
    int nchunks = bpc * prop.multiProcessorCount / qblocks;
    if (nchunks < 1) nchunks = 1;
    const int chunk_rows = (nt + nchunks - 1) / nchunks;
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
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_knn2, dim3(qblocks, nchunks, 1), dim3(256), 0, 0, q, nq, 0LL, t, nt, 0LL, chunk_rows, nchunks, 0, part, o,
                           o + nq, o + 2 * nq);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double clk = prop.clockRate * 1e3;
    printf("%s nt=%d chunks=%d x %d rows: %.3f ms  %.3e comparisons/s  %.1f SIMD-cycles per wave-row\n", VARIANT, nt, nchunks, chunk_rows,
           best, (double)nq * nt / (best * 1e-3), best * 1e-3 * clk * prop.multiProcessorCount * 4 / ((double)nq / 64 * nt));
    return 0;
}
