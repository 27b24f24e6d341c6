// How much vector-ALU work hides behind the integer matrix instructions of gfx950: cycles per
// v_mfma_i32_16x16x64_i8 / v_mfma_i32_32x32x32_i8 (four independent accumulation chains per wave, four waves per
// SIMD, every SIMD busy) with 0 .. 6 independent full-rate VALU instructions issued per matrix instruction.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <int NFILL, bool BIG>
__global__ __launch_bounds__(256) void k_probe(int iters, int* out) {
    i32x4 a = i32x4{(int)threadIdx.x, 1, 2, 3}, b = i32x4{4, 5, (int)blockIdx.x, 7};
    i32x4 acc[4];
    i32x16 big[2];
    uint32_t f[6] = {1u + threadIdx.x, 2, 3, 4, 5, 6};
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = i32x4{i, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) big[i][j] = i + j;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (BIG)
                big[i & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, big[i & 1], 0, 0, 0);
            else
                acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NFILL; k++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(f[k]) : "v"(f[(k + 1) % 6]));
        }
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 2; i++) s += big[i][0] + big[i][15];
#pragma unroll
    for (int k = 0; k < 6; k++) s += (int)f[k];
    if (s == 0x7fffffff) out[0] = s;
}
template <int NFILL, bool BIG>
static void run(const hipDeviceProp_t& prop, int* o) {
    const int iters = 20000, waves = 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_probe<NFILL, BIG>), dim3(prop.multiProcessorCount * waves), dim3(256), 0, 0, iters, o);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double clk = prop.clockRate * 1e3;
    printf("%s + %d v_xor per matrix instruction: %.1f cycles per matrix instruction (at the nominal %.0f MHz)\n",
           BIG ? "v_mfma_i32_32x32x32_i8" : "v_mfma_i32_16x16x64_i8", NFILL, best * 1e-3 * clk / ((double)iters * 4 * waves), clk / 1e6);
}
int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    int* o;
    (void)hipMalloc(&o, 64);
    run<0, false>(prop, o);
    run<2, false>(prop, o);
    run<4, false>(prop, o);
    run<6, false>(prop, o);
    run<0, true>(prop, o);
    run<2, true>(prop, o);
    run<4, true>(prop, o);
    run<6, true>(prop, o);
    return 0;
}
