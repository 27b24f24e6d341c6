// What does v_cvt_pk_u8_f32 do with fractions, ties and out-of-range values?  (GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int n) {
    int i = threadIdx.x;
    if (i < n) {
        unsigned r;
        asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %2" : "=v"(r) : "v"(in[i]), "v"(0xaabbccddu));
        out[i] = r;
    }
}
int main() {
    const float h[] = {0.0f, 0.49f, 0.5f, 0.51f, 0.99f, 1.0f, 1.5f, 2.5f, 3.5f, 2.999f, 254.5f, 254.99f, 255.0f, 255.5f, 255.99f, 256.0f, 257.9f, 1000.f, -0.5f, -3.f};
    const int n = sizeof(h) / sizeof(h[0]);
    float* di; unsigned* d_o; unsigned ho[32];
    hipMalloc(&di, sizeof(h)); hipMalloc(&d_o, 4 * 32);
    hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, d_o, n);
    hipMemcpy(ho, d_o, 4 * n, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("%9.3f -> %08x (byte %u)\n", h[i], ho[i], (ho[i] >> 8) & 255);
    return 0;
}
