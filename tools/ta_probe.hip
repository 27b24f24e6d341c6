// Cost model of the vector memory path (TA / TCP) of gfx950 for gather-like loads: how many vector-cache accesses and
// how much time one wave-level load instruction costs, by the shape of the addresses of its 64 lanes.  Every mode is
// a kernel of its own name, so that rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD reports them apart:
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ta_probe tools/ta_probe.hip && tools/bin/ta_probe
// The buffer is 8 MB of an "image" with rows of PITCH bytes (L2-resident after the first pass); every wave reads
// windows at pseudo-random positions, ITERS instructions per wave, results folded into one word so that nothing is
// optimised away.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define PITCH 704
#define ROWS 11000
#define ITERS 256
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__device__ __forceinline__ uint32_t lane_offset(int lane, uint32_t base) {
    // byte offset of the lane's access relative to `base` (a window corner)
    switch (MODE) {
        case 0: return 4u * lane;                                              // 256 contiguous aligned bytes
        case 1: return (lane >> 4) * PITCH + 4u * (lane & 15);                  // 4 rows x 64 B, aligned
        case 2: return (lane >> 3) * PITCH + 4u * (lane & 7);                   // 8 rows x 32 B, aligned
        case 3: return (lane >> 3) * PITCH + 4u * (lane & 7) + 1u;              // 8 rows x 32 B, dwords off by one byte
        case 4: return (lane / 11) * PITCH + 4u * (lane % 11);                  // rows of 11 dwords (44 B), aligned
        case 5: return (lane >> 2) * PITCH + 16u * (lane & 3);                  // 16 rows x 64 B as 16-byte pieces
        case 6: return (lane >> 2) * PITCH + 4u * (lane & 3);                   // 16 rows x 16 B
        case 7: return (lane >> 5) * PITCH + 4u * (lane & 31);                  // 2 rows x 128 B
        case 8: return (lane >> 1) * PITCH + 16u * (lane & 1);                  // 32 rows x 32 B as 16-byte pieces
        case 9: return (lane >> 2) * PITCH + 8u * (lane & 3);                   // 16 rows x 32 B as 8-byte pieces
        case 10: return (lane / 12) * PITCH + 4u * (lane % 12);                 // rows of 12 dwords
        case 11: case 12: case 14: return (lane >> 4) * PITCH + 4u * (lane & 15);  // rows of 16 lanes, 11 / 12 / 9 of them active
        case 13: return (lane >> 3) * PITCH + 4u * (lane & 7);                  // 8 rows, 8-byte loads 4 bytes apart
        case 15: return (lane / 11) * PITCH + 4u * (lane % 11);                 // rows of 11 dwords (16-byte aligned start)
        case 16: case 17: return (lane >> 2) * PITCH + 16u * (lane & 3);        // 16 rows of 4 x 16-byte loads (4 / 3 active)
        case 18: return (lane / 3) * PITCH + 16u * (lane % 3);                  // 21 rows of 3 x 16-byte loads
        case 19: return (lane / 5) * PITCH + 8u * (lane % 5);                   // 12 rows of 5 x 8-byte loads
        case 20: return (lane / 6) * PITCH + 8u * (lane % 6);                   // 10 rows of 6 x 8-byte loads
    }
    return 0;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(const uint8_t* __restrict__ img, uint32_t* __restrict__ out, int align_mask, int region_rows) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t seed = wave * 2654435761u + 12345u;
    uint32_t acc = 0;
    const uint32_t lo = lane_offset<MODE>(lane, 0);
    const int sub = MODE == 16 || MODE == 17 ? (lane & 3) : (lane & 15);
    const bool on = MODE == 11 ? sub < 11 : MODE == 12 ? sub < 12 : MODE == 14 ? sub < 9 : MODE == 17 ? sub < 3 : MODE == 18 ? lane < 63
                    : MODE == 19 || MODE == 20 ? lane < 60 : true;
#pragma unroll 8
    for (int it = 0; it < ITERS; it++) {
        seed = seed * 1664525u + 1013904223u;
        // window corner: a random row and column, wave-uniform; align_mask 63: 64-byte aligned columns, 3: dword aligned,
        // 0: any byte
        const uint32_t row = (seed >> 12) & (uint32_t)region_rows, col = ((seed >> 3) & 255u) & ~(uint32_t)align_mask;
        const uint8_t* p = img + (size_t)row * PITCH + col + lo;
        if (!on) continue;
        if (MODE == 5 || MODE == 8 || (MODE >= 16 && MODE <= 18)) {
            uint4 v;
            __builtin_memcpy(&v, p, 16);
            acc += v.x ^ v.y ^ v.z ^ v.w;
        } else if (MODE == 9 || MODE == 13 || MODE == 19 || MODE == 20) {
            uint2 v;
            __builtin_memcpy(&v, p, 8);
            acc += v.x ^ v.y;
        } else {
            uint32_t v;
            __builtin_memcpy(&v, p, 4);
            acc += v;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
static void run1(const uint8_t* d_img, uint32_t* d_out, int align_mask, const char* what, int region_rows) {
    const int blocks = 256 * 32;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), 0, 0, d_img, d_out, align_mask, region_rows);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), 0, 0, d_img, d_out, align_mask, region_rows);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = (double)blocks * 4 * ITERS;
    // cycles of one CU per wave instruction at 2.4 GHz with all 256 CUs busy
    printf("mode %2d align %2d rows %5d  %-44s %8.3f ms  %6.1f CU-cycles per wave instruction\n", MODE, align_mask + 1, region_rows, what, ms,
           ms * 1e-3 * 2.4e9 * 256 / instr);
}

template <int MODE>
static void run(const uint8_t* d_img, uint32_t* d_out, int align_mask, const char* what) {
    run1<MODE>(d_img, d_out, align_mask, what, 8191);
    run1<MODE>(d_img, d_out, align_mask, what, 7);  // windows of one 20 KB region: vector-cache hits
}

int main() {
    uint8_t* d_img;
    uint32_t* d_out;
    const size_t bytes = (size_t)PITCH * ROWS;
    CK(hipMalloc(&d_img, bytes + 4096));
    CK(hipMalloc(&d_out, 256 * 32 * 256 * 4));
    std::vector<uint8_t> h(bytes + 4096);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)(i * 131 + (i >> 8));
    CK(hipMemcpy(d_img, h.data(), h.size(), hipMemcpyHostToDevice));
    run<0>(d_img, d_out, 63, "256 contiguous bytes");
    run<1>(d_img, d_out, 63, "4 rows x 64 B");
    run<1>(d_img, d_out, 3, "4 rows x 64 B");
    run<2>(d_img, d_out, 63, "8 rows x 32 B");
    run<2>(d_img, d_out, 3, "8 rows x 32 B");
    run<3>(d_img, d_out, 3, "8 rows x 32 B, dwords off by one byte");
    run<4>(d_img, d_out, 3, "rows of 11 dwords");
    run<5>(d_img, d_out, 63, "16 rows x 64 B as 16-byte pieces");
    run<5>(d_img, d_out, 15, "16 rows x 64 B as 16-byte pieces");
    run<6>(d_img, d_out, 3, "16 rows x 16 B as dwords");
    run<7>(d_img, d_out, 3, "2 rows x 128 B");
    run<8>(d_img, d_out, 15, "32 rows x 32 B as 16-byte pieces");
    run<9>(d_img, d_out, 7, "16 rows x 32 B as 8-byte pieces");
    run<10>(d_img, d_out, 3, "rows of 12 dwords");
    run<11>(d_img, d_out, 3, "rows of 16 lanes, 11 active");
    run<12>(d_img, d_out, 3, "rows of 16 lanes, 12 active");
    run<14>(d_img, d_out, 3, "rows of 16 lanes, 9 active");
    run<13>(d_img, d_out, 3, "8 rows, 8-byte loads 4 bytes apart");
    run<15>(d_img, d_out, 15, "rows of 11 dwords, 16-byte aligned start");
    run<16>(d_img, d_out, 3, "16 rows of 4 x 16-byte loads, dword aligned");
    run<17>(d_img, d_out, 3, "16 rows of 3 x 16-byte loads, dword aligned");
    run<18>(d_img, d_out, 3, "21 rows of 3 x 16-byte loads, dword aligned");
    run<19>(d_img, d_out, 3, "12 rows of 5 x 8-byte loads, dword aligned");
    run<20>(d_img, d_out, 3, "10 rows of 6 x 8-byte loads, dword aligned");
    CK(hipDeviceSynchronize());
    return 0;
}
