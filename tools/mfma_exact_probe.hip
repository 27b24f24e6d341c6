// Does v_mfma_f32_32x32x64_f8f6f4 add small integer dot products to an accumulator input with 24 significant bits
// exactly?  Four chained instructions add sign * 4 * (m[row] + m[row + 32]) (|sum| <= 256, products of E2M1 0 / 1.0 with
// +-1.0) to C = base + k / 2^15; every result is representable in fp32 (|value| < 512, multiples of 2^-15), so any
// difference from the exact sum is rounding inside the matrix pipe.  Development probe for the Hamming scan, whose
// float keys use 19 of those bits today (orb-slam2_amd/csrc/knn2_mfma.h).
//   hipcc --offload-arch=gfx950 -O2 -o mfma_exact_probe tools/mfma_exact_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(64) void k_probe(const int* __restrict__ m, const float* __restrict__ c, float* __restrict__ out) {
    const int lane = threadIdx.x, col = lane & 31;
    const int trial = blockIdx.x;
    // m nibbles of 1.0 (0x2) in the lane's 32 K positions
    const int mm = m[trial * 64 + lane];
    uint32_t w[4];
    for (int k = 0; k < 4; k++) {
        const int n = min(max(mm - 8 * k, 0), 8);
        uint32_t v = 0;
        for (int j = 0; j < n; j++) v |= 0x2u << (4 * j);
        w[k] = v;
    }
    const i32x8 a = i32x8{(int)w[0], (int)w[1], (int)w[2], (int)w[3], 0, 0, 0, 0};
    const uint32_t bq = (col & 1) ? 0xAAAAAAAAu : 0x22222222u;  // -1 for odd columns
    const i32x8 b = i32x8{(int)bq, (int)bq, (int)bq, (int)bq, 0, 0, 0, 0};
    f32x16 acc;
    for (int i = 0; i < 16; i++) acc[i] = c[(trial * 64 + lane) * 16 + i];
    for (int k = 0; k < 4; k++) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, 0, 0, 0);
    for (int i = 0; i < 16; i++) out[(trial * 64 + lane) * 16 + i] = acc[i];
}

int main() {
    const int trials = 4096;
    std::vector<int> m(trials * 64);
    std::vector<float> c(trials * 64 * 16), out(trials * 64 * 16);
    uint64_t s = 88172645463325252ULL;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); };
    for (auto& x : m) x = rnd() % 33;
    for (int t = 0; t < trials; t++)
        for (int l = 0; l < 64; l++)
            for (int i = 0; i < 16; i++) {
                const int col = l & 31;
                const int base = (t & 1) ? 0 : ((col & 1) ? 256 : 255);  // keeps |C + sum| < 512 on both signs
                c[(t * 64 + l) * 16 + i] = (float)base + (float)(rnd() & 32767) / 32768.f;
            }
    int *dm; float *dc, *dout;
    if (hipMalloc(&dm, m.size() * 4) != hipSuccess || hipMalloc(&dc, c.size() * 4) != hipSuccess || hipMalloc(&dout, out.size() * 4) != hipSuccess) return 2;
    hipMemcpy(dm, m.data(), m.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(trials), dim3(64), 0, 0, dm, dc, dout);
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    long long bad = 0, total = 0, wide = 0;
    double worst = 0;
    for (int t = 0; t < trials; t++)
        for (int l = 0; l < 64; l++)
            for (int i = 0; i < 16; i++) {
                const int h = l >> 5, col = l & 31, row = (i & 3) + 8 * (i >> 2) + 4 * h;
                const int dot = 4 * (m[t * 64 + row] + m[t * 64 + row + 32]) * ((col & 1) ? -1 : 1);
                const double expect = (double)c[(t * 64 + l) * 16 + i] + dot;
                const float got = out[(t * 64 + l) * 16 + i];
                total++;
                if (expect >= 256 || expect <= -256) wide++;
                if ((double)got != expect) {
                    bad++;
                    if (fabs((double)got - expect) > worst) worst = fabs((double)got - expect);
                }
            }
    printf("results %lld (of which %lld with |value| >= 256, i.e. all 24 bits in use), inexact %lld, worst error %.3g (2^-15 = %.3g)\n",
           total, wide, bad, worst, 1.0 / 32768);
    return 0;
}
