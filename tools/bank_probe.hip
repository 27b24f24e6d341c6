// VGPR bank probe: does a wave64 VALU instruction whose two vector sources sit a given distance apart in
// the register file issue slower?  Physical registers are named explicitly; v20.. are clobbered.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define KERNEL(name, body)                                                                   \
    __global__ __launch_bounds__(256) void name(uint32_t* out, int iters) {                  \
        for (int i = 0; i < iters; i++) {                                                    \
            asm volatile(REP8(body) ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", \
                         "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47"); \
        }                                                                                    \
        if (iters < 0) out[0] = 1;                                                           \
    }
// eight independent instructions per body; sources v20 + {0,1,2,4,8,16}
#define BODY(op, a, b)                                                                             \
    op " v40, " a ", " b "\n" op " v41, " a ", " b "\n" op " v42, " a ", " b "\n" op " v43, " a ", " b "\n" \
    op " v44, " a ", " b "\n" op " v45, " a ", " b "\n" op " v46, " a ", " b "\n" op " v47, " a ", " b "\n"
KERNEL(xor_d0, BODY("v_xor_b32", "v20", "v20"))
KERNEL(xor_d1, BODY("v_xor_b32", "v20", "v21"))
KERNEL(xor_d2, BODY("v_xor_b32", "v20", "v22"))
KERNEL(xor_d4, BODY("v_xor_b32", "v20", "v24"))
KERNEL(xor_d8, BODY("v_xor_b32", "v20", "v28"))
KERNEL(xor_d16, BODY("v_xor_b32", "v20", "v36"))
KERNEL(bcnt_d1, BODY("v_bcnt_u32_b32", "v20", "v21"))
KERNEL(bcnt_d2, BODY("v_bcnt_u32_b32", "v20", "v22"))
KERNEL(bcnt_d4, BODY("v_bcnt_u32_b32", "v20", "v24"))
KERNEL(bcnt_d8, BODY("v_bcnt_u32_b32", "v20", "v28"))
// destination bank vs source bank: dst v40.. (bank 0 if 4 banks) with sources v21, v22
#define BODYD(op, d, a, b) op " " d ", " a ", " b "\n" op " " d ", " a ", " b "\n" op " " d ", " a ", " b "\n" op " " d ", " a ", " b "\n" \
    op " " d ", " a ", " b "\n" op " " d ", " a ", " b "\n" op " " d ", " a ", " b "\n" op " " d ", " a ", " b "\n"
KERNEL(xor_dst_same_a, BODYD("v_xor_b32", "v40", "v20", "v21"))
KERNEL(xor_dst_other, BODYD("v_xor_b32", "v41", "v20", "v22"))
// the matcher's pairing: xor of a fresh temp then bcnt accumulating
KERNEL(mix_conf, "v_xor_b32 v40, v20, v24\n v_bcnt_u32_b32 v44, v40, v44\n v_xor_b32 v41, v21, v25\n v_bcnt_u32_b32 v45, v41, v45\n"
                 "v_xor_b32 v42, v22, v26\n v_bcnt_u32_b32 v46, v42, v46\n v_xor_b32 v43, v23, v27\n v_bcnt_u32_b32 v47, v43, v47\n")
KERNEL(mix_free, "v_xor_b32 v40, v20, v25\n v_bcnt_u32_b32 v45, v40, v45\n v_xor_b32 v41, v21, v26\n v_bcnt_u32_b32 v46, v41, v46\n"
                 "v_xor_b32 v42, v22, v27\n v_bcnt_u32_b32 v47, v42, v47\n v_xor_b32 v43, v23, v24\n v_bcnt_u32_b32 v44, v43, v44\n")

// independent alternation: the popcounts do not read the xor results
KERNEL(mix_indep, "v_xor_b32 v40, v20, v25\n v_bcnt_u32_b32 v45, v21, v45\n v_xor_b32 v41, v21, v26\n v_bcnt_u32_b32 v46, v22, v46\n"
                  "v_xor_b32 v42, v22, v27\n v_bcnt_u32_b32 v47, v23, v47\n v_xor_b32 v43, v23, v24\n v_bcnt_u32_b32 v44, v20, v44\n")
// four xors, then the four popcounts that read them (distance 4 between producer and consumer)
KERNEL(grp4_dep, "v_xor_b32 v40, v20, v25\n v_xor_b32 v41, v21, v26\n v_xor_b32 v42, v22, v27\n v_xor_b32 v43, v23, v24\n"
                 "v_bcnt_u32_b32 v45, v40, v45\n v_bcnt_u32_b32 v46, v41, v46\n v_bcnt_u32_b32 v47, v42, v47\n v_bcnt_u32_b32 v44, v43, v44\n")
// same, one accumulator chain (each popcount reads the previous one's result)
KERNEL(grp4_chain, "v_xor_b32 v40, v20, v25\n v_xor_b32 v41, v21, v26\n v_xor_b32 v42, v22, v27\n v_xor_b32 v43, v23, v24\n"
                   "v_bcnt_u32_b32 v44, v40, v44\n v_bcnt_u32_b32 v44, v41, v44\n v_bcnt_u32_b32 v44, v42, v44\n v_bcnt_u32_b32 v44, v43, v44\n")
// two xors, two popcounts
KERNEL(grp2_dep, "v_xor_b32 v40, v20, v25\n v_xor_b32 v41, v21, v26\n v_bcnt_u32_b32 v45, v40, v45\n v_bcnt_u32_b32 v46, v41, v46\n"
                 "v_xor_b32 v42, v22, v27\n v_xor_b32 v43, v23, v24\n v_bcnt_u32_b32 v47, v42, v47\n v_bcnt_u32_b32 v44, v43, v44\n")
// xor whose second source was just written by the previous xor
KERNEL(xor_chain, "v_xor_b32 v40, v40, v25\n v_xor_b32 v40, v40, v26\n v_xor_b32 v40, v40, v27\n v_xor_b32 v40, v40, v24\n"
                  "v_xor_b32 v40, v40, v25\n v_xor_b32 v40, v40, v26\n v_xor_b32 v40, v40, v27\n v_xor_b32 v40, v40, v24\n")
KERNEL(bcnt_chain, "v_bcnt_u32_b32 v40, v20, v40\n v_bcnt_u32_b32 v40, v21, v40\n v_bcnt_u32_b32 v40, v22, v40\n v_bcnt_u32_b32 v40, v23, v40\n"
                   "v_bcnt_u32_b32 v40, v20, v40\n v_bcnt_u32_b32 v40, v21, v40\n v_bcnt_u32_b32 v40, v22, v40\n v_bcnt_u32_b32 v40, v23, v40\n")
// the filter's extras
KERNEL(or_only, BODY("v_or_b32", "v20", "v21"))
KERNEL(or3_only, "v_or3_b32 v40, v20, v21, v22\n v_or3_b32 v41, v20, v21, v22\n v_or3_b32 v42, v20, v21, v22\n v_or3_b32 v43, v20, v21, v22\n"
                 "v_or3_b32 v44, v20, v21, v22\n v_or3_b32 v45, v20, v21, v22\n v_or3_b32 v46, v20, v21, v22\n v_or3_b32 v47, v20, v21, v22\n")
KERNEL(cmp_only, "v_cmp_gt_i32 vcc, 0, v20\n v_cmp_gt_i32 vcc, 0, v21\n v_cmp_gt_i32 vcc, 0, v22\n v_cmp_gt_i32 vcc, 0, v23\n"
                 "v_cmp_gt_i32 vcc, 0, v20\n v_cmp_gt_i32 vcc, 0, v21\n v_cmp_gt_i32 vcc, 0, v22\n v_cmp_gt_i32 vcc, 0, v23\n")

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
    typedef void (*K)(uint32_t*, int);
    auto run = [&](const char* name, K k) {
        const int blocks = cus * 8, iters = 2048;  // 8 waves per SIMD resident, one round
        float best = 1e9f;
        for (int r = 0; r < 4; r++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double insts = 8.0 * iters * 64;  // wave instructions per wave
        printf("%-16s %7.3f ms  %.2f SIMD-cycles per wave instruction\n", name, best, best * 1e-3 * clk / (insts * 8) * 8);
    };
#define R(k) run(#k, k)
    R(xor_d0); R(xor_d1); R(xor_d2); R(xor_d4); R(xor_d8); R(xor_d16);
    R(bcnt_d1); R(bcnt_d2); R(bcnt_d4); R(bcnt_d8);
    R(xor_dst_same_a); R(xor_dst_other); R(mix_conf); R(mix_free);
    R(mix_indep); R(grp4_dep); R(grp4_chain); R(grp2_dep); R(xor_chain); R(bcnt_chain); R(or_only); R(or3_only); R(cmp_only);
    return 0;
}
