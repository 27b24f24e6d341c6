// Bandwidth probe: what copy rate do the extractor's access patterns reach on this device?
//   A  (64,4) blocks, one dword per lane per row, each thread walks 32 rows of a strided plane
//   B  the same with 16 bytes per lane
//   C  flat grid-stride dword copy      D  flat grid-stride uint4 copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <class T>
__global__ __launch_bounds__(256) void k_rows(const uint8_t* src, uint8_t* dst, int stride, int ph, long long frame_bytes) {
    const int x = (blockIdx.x * 64 + threadIdx.x) * (int)sizeof(T);
    const int y0 = (blockIdx.y * 4 + threadIdx.y) * 32;
    if (x >= stride || y0 >= ph) return;
    const long long base = (long long)blockIdx.z * frame_bytes;
#pragma unroll 8
    for (int i = 0; i < 32; i++) {
        const int y = y0 + i;
        if (y < ph) *(T*)(dst + base + (long long)y * stride + x) = *(const T*)(src + base + (long long)y * stride + x);
    }
}
template <class T>
__global__ __launch_bounds__(256) void k_flat(const T* src, T* dst, long long n) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) dst[i] = src[i];
}

int main() {
    const int stride = 1344, ph = 752, frames = 512;
    const long long fb = (long long)stride * ph, total = fb * frames;
    uint8_t *a, *b;
    hipMalloc(&a, total); hipMalloc(&b, total);
    hipMemset(a, 1, total); hipMemset(b, 2, total);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch) {
        float best = 1e9;
        for (int r = 0; r < 5; r++) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%s: %.3f ms  %.2f TB/s (read+write)\n", name, best, 2.0 * total / best / 1e9);
    };
    timeit("A rows dword ", [&] { hipLaunchKernelGGL(k_rows<uint32_t>, dim3((stride / 4 + 63) / 64, (ph + 127) / 128, frames), dim3(64, 4), 0, 0, a, b, stride, ph, fb); });
    timeit("B rows uint4 ", [&] { hipLaunchKernelGGL(k_rows<uint4>, dim3((stride / 16 + 63) / 64, (ph + 127) / 128, frames), dim3(64, 4), 0, 0, a, b, stride, ph, fb); });
    timeit("C flat dword ", [&] { hipLaunchKernelGGL(k_flat<uint32_t>, dim3(256 * 32), dim3(256), 0, 0, (const uint32_t*)a, (uint32_t*)b, total / 4); });
    timeit("D flat uint4 ", [&] { hipLaunchKernelGGL(k_flat<uint4>, dim3(256 * 32), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, total / 16); });
    return 0;
}
