// Where do the workgroups of a grid that fits the chip in one round land?  Every workgroup (256 threads,
// 16 KB of LDS, ~80 VGPRs) records the XCC / SE / CU it runs on and spins long enough for all to be co-resident.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k_where(uint32_t* out, int spin) {
    __shared__ uint32_t pad[4096];
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    uint32_t acc = pad[(threadIdx.x * 7) & 4095];
    for (int i = 0; i < spin; i++) asm volatile("v_bcnt_u32_b32 %0, %0, %0\n s_nop 0" : "+v"(acc));
    if (threadIdx.x == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 2] = hw;
        out[blockIdx.x * 2 + 1] = xcc | (acc == 0x12345 ? 1u << 31 : 0);
    }
}
int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 1024;
    uint32_t* d;
    hipMalloc(&d, blocks * 8);
    hipLaunchKernelGGL(k_where, dim3(blocks), dim3(256), 0, 0, d, 20000);
    std::vector<uint32_t> h(blocks * 2);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    std::map<uint32_t, int> per_cu;
    for (int b = 0; b < blocks; b++) {
        const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
    }
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second]++;
    printf("%d workgroups on %zu distinct CUs; workgroups per CU -> number of CUs:", blocks, per_cu.size());
    for (auto& kv : hist) printf("  %d -> %d", kv.first, kv.second);
    printf("\n");
    return 0;
}
