// Issue-rate probe: how many wave64 instructions per second does the device sustain for the
// integer/packed/float VALU operations the extractor and matcher kernels are built from?
// Every kernel runs 8 independent dependency chains per lane, fully unrolled, at full occupancy.
// Output: wave-instructions/s and the implied SIMD cycles per instruction at the reported clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 256
#define CHAINS 8

#define PROBE2(name, INSN)                                                                          \
    __global__ __launch_bounds__(256) void k_##name(uint32_t* out, uint32_t seed) {                 \
        uint32_t r[CHAINS];                                                                         \
        const uint32_t b = seed ^ threadIdx.x;                                                      \
        for (int c = 0; c < CHAINS; c++) r[c] = seed * (c + 1) + threadIdx.x;                       \
        for (int i = 0; i < ITERS; i++) {                                                           \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++)                                      \
                asm volatile(INSN " %0, %1, %2" : "=v"(r[c]) : "v"(r[c]), "v"(b));                  \
        }                                                                                           \
        uint32_t s = 0;                                                                             \
        for (int c = 0; c < CHAINS; c++) s ^= r[c];                                                 \
        if (s == 0x12345678u) out[0] = s;                                                           \
    }
#define PROBE3(name, INSN)                                                                          \
    __global__ __launch_bounds__(256) void k_##name(uint32_t* out, uint32_t seed) {                 \
        uint32_t r[CHAINS];                                                                         \
        const uint32_t b = seed ^ threadIdx.x, c3 = seed + 7u * threadIdx.x;                        \
        for (int c = 0; c < CHAINS; c++) r[c] = seed * (c + 1) + threadIdx.x;                       \
        for (int i = 0; i < ITERS; i++) {                                                           \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++)                                      \
                asm volatile(INSN " %0, %1, %2, %3" : "=v"(r[c]) : "v"(r[c]), "v"(b), "v"(c3));     \
        }                                                                                           \
        uint32_t s = 0;                                                                             \
        for (int c = 0; c < CHAINS; c++) s ^= r[c];                                                 \
        if (s == 0x12345678u) out[0] = s;                                                           \
    }
#define PROBE1(name, INSN)                                                                          \
    __global__ __launch_bounds__(256) void k_##name(uint32_t* out, uint32_t seed) {                 \
        uint32_t r[CHAINS];                                                                         \
        for (int c = 0; c < CHAINS; c++) r[c] = seed * (c + 1) + threadIdx.x;                       \
        for (int i = 0; i < ITERS; i++) {                                                           \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++)                                      \
                asm volatile(INSN " %0, %1" : "=v"(r[c]) : "v"(r[c]));                              \
        }                                                                                           \
        uint32_t s = 0;                                                                             \
        for (int c = 0; c < CHAINS; c++) s ^= r[c];                                                 \
        if (s == 0x12345678u) out[0] = s;                                                           \
    }
// 64-bit (register pair) operations
#define PROBE2D(name, INSN)                                                                         \
    __global__ __launch_bounds__(256) void k_##name(uint32_t* out, uint32_t seed) {                 \
        double r[CHAINS];                                                                           \
        const double b = 1.0 + 1e-9 * threadIdx.x;                                                  \
        for (int c = 0; c < CHAINS; c++) r[c] = 1.0 + seed * 1e-9 * (c + 1);                        \
        for (int i = 0; i < ITERS; i++) {                                                           \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++)                                      \
                asm volatile(INSN " %0, %1, %2" : "=v"(r[c]) : "v"(r[c]), "v"(b));                  \
        }                                                                                           \
        double s = 0;                                                                               \
        for (int c = 0; c < CHAINS; c++) s += r[c];                                                 \
        if (s == 0.12345) out[0] = 1;                                                               \
    }
// LDS reads: every lane reads its own address (conflict-free dwords / 4 lanes per dword for bytes)
#define PROBELDS(name, INSN, SCALE)                                                                 \
    __global__ __launch_bounds__(256) void k_##name(uint32_t* out, uint32_t seed) {                 \
        __shared__ uint32_t lds[4096];                                                              \
        for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed + i;                            \
        __syncthreads();                                                                            \
        uint32_t r[CHAINS];                                                                         \
        const uint32_t a = (uint32_t)(uintptr_t)lds + threadIdx.x * SCALE;                          \
        for (int c = 0; c < CHAINS; c++) r[c] = 0;                                                  \
        for (int i = 0; i < ITERS; i++) {                                                           \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++)                                      \
                asm volatile(INSN " %0, %1 offset:%2" : "=v"(r[c]) : "v"(a), "n"(c * 1024));        \
            asm volatile("s_waitcnt lgkmcnt(0)");                                                   \
        }                                                                                           \
        uint32_t s = 0;                                                                             \
        for (int c = 0; c < CHAINS; c++) s ^= r[c];                                                 \
        if (s == 0x12345678u) out[0] = s;                                                           \
    }

#define PROBESDWA(name, TEXT)                                                                       \
    __global__ __launch_bounds__(256) void k_##name(uint32_t* out, uint32_t seed) {                 \
        uint32_t r[CHAINS];                                                                         \
        const uint32_t b = seed ^ threadIdx.x;                                                      \
        for (int c = 0; c < CHAINS; c++) r[c] = seed * (c + 1) + threadIdx.x;                       \
        for (int i = 0; i < ITERS; i++) {                                                           \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++)                                      \
                asm volatile(TEXT : "+v"(r[c]) : "v"(b));                                           \
        }                                                                                           \
        uint32_t s = 0;                                                                             \
        for (int c = 0; c < CHAINS; c++) s ^= r[c];                                                 \
        if (s == 0x12345678u) out[0] = s;                                                           \
    }
PROBESDWA(add_sdwa_w1, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1")
PROBESDWA(lshr_sdwa_b1, "v_lshrrev_b32_sdwa %0, %1, %0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD")
PROBESDWA(add_dpp, "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
PROBE2(add_u32, "v_add_u32")
PROBE2(sub_u32, "v_sub_u32")
PROBE2(xor_b32, "v_xor_b32")
PROBE2(and_b32, "v_and_b32")
PROBE2(min_i32, "v_min_i32")
PROBE2(max_u32, "v_max_u32")
PROBE2(lshrrev, "v_lshrrev_b32")
PROBE2(mul_lo_u32, "v_mul_lo_u32")
PROBE2(mul_u32_u24, "v_mul_u32_u24")
PROBE2(pk_sub_i16, "v_pk_sub_i16")
PROBE2(pk_min_i16, "v_pk_min_i16")
PROBE2(pk_max_i16, "v_pk_max_i16")
PROBE2(pk_add_u16, "v_pk_add_u16")
PROBE2(pk_mul_lo_u16, "v_pk_mul_lo_u16")
PROBE2(bcnt, "v_bcnt_u32_b32")
PROBE2(mul_hi_u32, "v_mul_hi_u32")
PROBE2(mul_hi_u32_u24, "v_mul_hi_u32_u24")
PROBE2(lshlrev, "v_lshlrev_b32")
PROBE2(or_b32, "v_or_b32")
PROBE2(ashrrev, "v_ashrrev_i32")
PROBE2(cndmask, "v_cndmask_b32")
PROBE2(add_f16, "v_add_f16")
PROBE2(pk_fma_dummy, "v_pk_lshrrev_b16")
PROBE2(mul_f32, "v_mul_f32")
PROBE2(add_f32, "v_add_f32")
PROBE2(max_f32, "v_max_f32")
PROBE2(mbcnt_lo, "v_mbcnt_lo_u32_b32")
PROBE3(min3_i32, "v_min3_i32")
PROBE3(max3_i32, "v_max3_i32")
PROBE3(med3_i32, "v_med3_i32")
PROBE3(perm_b32, "v_perm_b32")
PROBE3(mad_i32_i24, "v_mad_i32_i24")
PROBE3(mad_u32_u24, "v_mad_u32_u24")
PROBE3(lshl_add_u32, "v_lshl_add_u32")
PROBE3(add3_u32, "v_add3_u32")
PROBE3(and_or_b32, "v_and_or_b32")
PROBE3(lshl_or_b32, "v_lshl_or_b32")
PROBE3(bfe_u32, "v_bfe_u32")
PROBE3(alignbyte, "v_alignbyte_b32")
PROBE3(dot4_u32_u8, "v_dot4_u32_u8")
PROBE3(dot2_u32_u16, "v_dot2_u32_u16")
PROBE3(sad_u8, "v_sad_u8")
PROBE3(sad_u32, "v_sad_u32")
PROBE3(fma_f32, "v_fma_f32")
PROBE3(xad_u32, "v_xad_u32")
PROBE3(mad_u32_u16, "v_mad_u32_u16")
PROBE3(add_lshl, "v_add_lshl_u32")
PROBE3(or3, "v_or3_b32")
PROBE3(min3_f32, "v_min3_f32")
PROBE3(pk_mad_u16, "v_pk_mad_u16")
PROBE3(cvt_pk_u8_f32, "v_cvt_pk_u8_f32")
PROBE2(max_u16, "v_max_u16")
PROBE2(min_i16, "v_min_i16")
PROBE2(max_f16, "v_max_f16")
PROBE2(min_f16, "v_min_f16")
PROBE2(sub_u16, "v_sub_u16")
PROBE2(pk_max_f16, "v_pk_max_f16")
PROBE2(pk_min_f16, "v_pk_min_f16")
PROBE2(pk_add_f16, "v_pk_add_f16")
PROBE3(min3_f16, "v_min3_f16")
PROBE3(max3_f16, "v_max3_f16")
PROBE3(min3_u16, "v_min3_u16")
PROBE3(max3_i16, "v_max3_i16")
PROBE3(med3_f16, "v_med3_f16")
PROBE3(max3_u32, "v_max3_u32")
PROBE1(mov_b32, "v_mov_b32")
PROBE1(cvt_f32_i32, "v_cvt_f32_i32")
PROBE1(cvt_f32_ubyte0, "v_cvt_f32_ubyte0")
PROBE1(rndne_f32, "v_rndne_f32")
PROBE1(rcp_f32, "v_rcp_f32")
PROBE1(ffbl, "v_ffbl_b32")
PROBE2D(mul_f64, "v_mul_f64")
PROBE2D(add_f64, "v_add_f64")
PROBE2D(pk_mul_f32, "v_pk_mul_f32")
PROBE2D(pk_add_f32, "v_pk_add_f32")
PROBELDS(ds_read_b32, "ds_read_b32", 4)
PROBELDS(ds_read_u8, "ds_read_u8", 1)
PROBELDS(ds_read_u8_spread, "ds_read_u8", 5)
PROBELDS(ds_read_u16, "ds_read_u16", 2)

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double clk = prop.clockRate * 1e3;  // Hz
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    uint32_t* out;
    hipMalloc(&out, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = cus * 8 * 4;
    auto run = [&](const char* name, void (*k)(uint32_t*, uint32_t)) {
        float best = 1e9f;
        for (int r = 0; r < 4; r++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 12345u + r);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double winstr = (double)blocks * 4 * ITERS * CHAINS;
        const double rate = winstr / (best * 1e-3);
        printf("%-22s %8.3f ms  %.3e wave-instr/s  %.2f SIMD-cycles/instr\n", name, best, rate, cus * 4.0 * clk / rate);
    };
#define RUN(name) run(#name, k_##name)
    RUN(add_sdwa_w1); RUN(lshr_sdwa_b1); RUN(add_dpp); RUN(add_u32); RUN(sub_u32); RUN(xor_b32); RUN(and_b32); RUN(min_i32); RUN(max_u32); RUN(lshrrev);
    RUN(mul_lo_u32); RUN(mul_u32_u24); RUN(pk_sub_i16); RUN(pk_min_i16); RUN(pk_max_i16); RUN(pk_add_u16);
    RUN(pk_mul_lo_u16); RUN(bcnt); RUN(mul_f32); RUN(add_f32); RUN(mbcnt_lo);
    RUN(min3_i32); RUN(max3_i32); RUN(med3_i32); RUN(perm_b32); RUN(mad_i32_i24); RUN(mad_u32_u24);
    RUN(lshl_add_u32); RUN(add3_u32); RUN(and_or_b32); RUN(lshl_or_b32); RUN(bfe_u32); RUN(alignbyte);
    RUN(dot4_u32_u8); RUN(dot2_u32_u16); RUN(sad_u8); RUN(sad_u32); RUN(fma_f32); RUN(xad_u32); RUN(max_f32); RUN(mul_hi_u32); RUN(mul_hi_u32_u24); RUN(lshlrev); RUN(or_b32); RUN(ashrrev); RUN(cndmask); RUN(add_f16); RUN(pk_fma_dummy); RUN(mad_u32_u16); RUN(add_lshl); RUN(or3); RUN(min3_f32); RUN(pk_mad_u16); RUN(cvt_pk_u8_f32);
    RUN(max_u16); RUN(min_i16); RUN(max_f16); RUN(min_f16); RUN(sub_u16); RUN(pk_max_f16); RUN(pk_min_f16); RUN(pk_add_f16);
    RUN(min3_f16); RUN(max3_f16); RUN(min3_u16); RUN(max3_i16); RUN(med3_f16); RUN(max3_u32);
    RUN(mov_b32); RUN(cvt_f32_i32); RUN(cvt_f32_ubyte0); RUN(rndne_f32); RUN(rcp_f32); RUN(ffbl);
    RUN(mul_f64); RUN(add_f64); RUN(pk_mul_f32); RUN(pk_add_f32);
    RUN(ds_read_b32); RUN(ds_read_u8); RUN(ds_read_u8_spread); RUN(ds_read_u16);
    return 0;
}
