// Brute-force Hamming k=2 nearest-neighbour search over 256-bit ORB descriptors for gfx950:
// the best / second-best ascending scan that every Search* method of the reference runs
// (src/ORBmatcher.cc:461-485) with DescriptorDistance (:1805-1821) as the metric, evaluated
// over the full train set of each (query set, train set) pair.
//
// Mapping: one lane owns one query (its 8 words live in VGPRs); the train rows are
// wave-uniform, so they arrive through scalar loads and feed v_xor_b32 / v_bcnt_u32_b32 as SGPR
// operands — no LDS, no MFMA (this is integer popcount work, not a dense contraction).
// A candidate is folded into the running top-2 as one packed key (distance << 22 | row), which
// makes "first index attaining the minimum" and "second order statistic with multiplicity" fall
// out of two unsigned min/max operations in ascending row order.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <algorithm>
#include <climits>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/orbhip.h"
#include "common.h"
#include "matcher_internal.h"

namespace {

#include "knn2_kernel.h"
}  // namespace
#include "knn2_merge.h"
namespace {

// Chunks are index-contiguous and visited in ascending order, so folding each chunk's two
// smallest distances with the reference's update rule reproduces the serial scan.
__global__ __launch_bounds__(256) void k_knn2_fold(const uint2* __restrict__ part, int nq, int nchunks, int chunk_rows,
                                                   int idx_base, int32_t* __restrict__ o_idx, int32_t* __restrict__ o_d1,
                                                   int32_t* __restrict__ o_d2) {
    const int pair = blockIdx.y;
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    int best = INT_MAX, best2 = INT_MAX, idx = -1;
    // the loads of eight chunks are issued together (they do not depend on the fold), then folded in order
    for (int c0 = 0; c0 < nchunks; c0 += 8) {
        uint2 pv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) pv[u] = part[((long long)pair * nchunks + min(c0 + u, nchunks - 1)) * nq + qi];
#pragma unroll
        for (int u = 0; u < 8; u++) {
        const int c = c0 + u;
        const uint2 p = pv[u];
        if (c >= nchunks || p.x == KNN_NONE) continue;
        const int d1 = (int)(p.x >> KNN_IDX_BITS);
        const int d2 = p.y == KNN_NONE ? INT_MAX : (int)(p.y >> KNN_IDX_BITS);
        if (d1 < best) {
            best2 = best;
            best = d1;
            idx = c * chunk_rows + (int)(p.x & KNN_IDX_MASK) + idx_base;
        } else if (d1 < best2) {
            best2 = d1;
        }
        if (d2 < best2) best2 = d2;
        }
    }
    const long long o = (long long)pair * nq + qi;
    o_idx[o] = idx;
    o_d1[o] = best;
    o_d2[o] = best2;
}

// Merge of per-shard (idx, d1, d2) triples of index-contiguous shards, ascending shard order.
__global__ __launch_bounds__(256) void k_knn2_merge(const int32_t* __restrict__ parts, int nshards, int nq,
                                                    int32_t* __restrict__ o_idx, int32_t* __restrict__ o_d1,
                                                    int32_t* __restrict__ o_d2) {
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    knn2_merge_query(parts, nshards, nq, qi, o_idx, o_d1, o_d2);
}

// src/ORBmatcher.cc:237 ((float)d1 < r*(float)d2), :487 (d1 <= d2*r, evaluated in float), with
// the distance gate d1 <= th.  One multiply and one compare: nothing to contract.
__global__ __launch_bounds__(256) void k_ratio(const int32_t* __restrict__ d1, const int32_t* __restrict__ d2, int n,
                                               float ratio, int form, int th, uint8_t* __restrict__ accept) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int a = d1[i], b = d2[i];
    bool ok;
    if (form == 0) ok = (float)a < __fmul_rn(ratio, (float)b);
    else ok = (float)a <= __fmul_rn((float)b, ratio);
    accept[i] = (ok && a <= th) ? 1 : 0;
}

// Micro-benchmarks of the integer VALU instructions the matcher is made of (sustained rate with
// every CU busy, 8 independent dependency chains per lane).  mode 0: v_bcnt_u32_b32 only,
// 1: v_xor_b32 + v_bcnt_u32_b32 pairs (the matcher's inner mix), 2: v_add_u32 only.
__global__ __launch_bounds__(256) void k_valu_rate(uint32_t* out, int iters, int mode, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t x = seed ^ (blockIdx.x * 2654435761u);
    if (mode == 0) {
        for (int i = 0; i < iters; i++) {
            asm volatile(
                "v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %8, %2\n v_bcnt_u32_b32 %3, %8, %3\n"
                "v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %8, %5\n v_bcnt_u32_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %8, %7\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                : "v"(x));
        }
    } else if (mode == 1) {
        uint32_t t0, t1, t2, t3;
        for (int i = 0; i < iters; i++) {
            asm volatile(
                "v_xor_b32 %8, %12, %0\n v_xor_b32 %9, %12, %1\n v_xor_b32 %10, %12, %2\n v_xor_b32 %11, %12, %3\n"
                "v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %9, %5\n v_bcnt_u32_b32 %6, %10, %6\n v_bcnt_u32_b32 %7, %11, %7\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(t0), "=&v"(t1),
                  "=&v"(t2), "=&v"(t3)
                : "v"(x));
        }
    } else {
        for (int i = 0; i < iters; i++) {
            asm volatile(
                "v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n"
                "v_add_u32 %4, %8, %4\n v_add_u32 %5, %8, %5\n v_add_u32 %6, %8, %6\n v_add_u32 %7, %8, %7\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                : "v"(x));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

}  // namespace

static int default_scan_kernel();

// Partial results of the chunked scan for the entry point that allocates nothing.  Every GPU the
// library runs on holds its own copy of this array at its own address until the process ends; host
// code reaches it only through the address hipGetSymbolAddress returns for the current device
// (the array's name is a host-side stand-in, not a device address).
#define KNN_SCRATCH_BYTES (16u << 20)
static __device__ __align__(256) unsigned char scratch[KNN_SCRATCH_BYTES];
#define KNN_SCRATCH_MAX_DEVICES 64
static void* g_scratch_ptr[KNN_SCRATCH_MAX_DEVICES];
// One event per device orders the calls that use the device's array against each other, whichever handle (and
// stream) they come from: a call makes its stream wait for the event, and records it again behind its last kernel.
static hipEvent_t g_scratch_event[KNN_SCRATCH_MAX_DEVICES];
static std::mutex g_scratch_mutex;

// Resolves, once per device and with that device current, the address of the device's scratch array.
static int scratch_for_current_device(int device, void** out) {
    *out = nullptr;
    if (device < 0 || device >= KNN_SCRATCH_MAX_DEVICES) return orb_fail(ORBHIP_EINVAL, "device ordinal %d out of range", device);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        const hipError_t err = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (err != hipSuccess || !ptr)
            return orb_fail(ORBHIP_EHIP, "hipGetSymbolAddress(scratch) failed: %s", hipGetErrorString(err));
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
            return orb_fail(ORBHIP_EHIP, "event creation for the scratch array failed");
        g_scratch_event[device] = ev;
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return ORBHIP_OK;
}

extern "C" {

int orbhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int orbhip_matcher_create(orbhip_matcher** out, int device) {
    if (!out) return orb_fail(ORBHIP_EINVAL, "null out pointer");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return orb_fail(ORBHIP_ENODEV, "no HIP device");
    if (device < 0 || device >= ndev) return orb_fail(ORBHIP_EINVAL, "device %d of %d", device, ndev);
    HIPCHK(hipSetDevice(device));
    orbhip_matcher* m = new orbhip_matcher;
    m->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) m->sm_count = prop.multiProcessorCount;
    m->scan_kernel = default_scan_kernel();
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&m->ev0) != hipSuccess ||
        hipEventCreate(&m->ev1) != hipSuccess || hipMalloc((void**)&m->d_err, 16) != hipSuccess ||
        hipMemset(m->d_err, 0, 16) != hipSuccess) {
        orbhip_matcher_destroy(m);
        return orb_fail(ORBHIP_EHIP, "stream/event creation failed");
    }
    {
        const int rc = scratch_for_current_device(device, &m->d_scratch);
        if (rc) {
            orbhip_matcher_destroy(m);
            return rc;
        }
    }
    *out = m;
    return ORBHIP_OK;
}

void orbhip_matcher_destroy(orbhip_matcher* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->d_part) (void)hipFree(m->d_part);
    if (m->d_stage) (void)hipFree(m->d_stage);
    if (m->d_gather) (void)hipFree(m->d_gather);
    if (m->d_work) (void)hipFree(m->d_work);
    if (m->d_topk) (void)hipFree(m->d_topk);
    if (m->d_err) (void)hipFree(m->d_err);
    if (m->ev0) (void)hipEventDestroy(m->ev0);
    if (m->ev1) (void)hipEventDestroy(m->ev1);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

int orbhip_matcher_synchronize(orbhip_matcher* m) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipStreamSynchronize(m->stream));
    if (hipEventElapsedTime(&m->last_ms, m->ev0, m->ev1) != hipSuccess) {
        m->last_ms = 0.f;
        (void)hipGetLastError();
    }
    struct {
        int flags, pad;
        long long need;
    } st = {0, 0, 0};
    HIPCHK(hipMemcpy(&st, m->d_err, sizeof(st), hipMemcpyDeviceToHost));
    if (st.flags) {
        (void)hipMemset(m->d_err, 0, sizeof(st));
        if (st.flags & 8)
            return orb_fail(ORBHIP_ECAP, "the candidate lists of a matcher call need %lld entries, more than the reservation "
                            "(orbhip_matcher_reserve_candidates); that call produced no matches", st.need);
        return orb_fail(ORBHIP_ECAP, "device output capacity exceeded (flags %d)", st.flags);
    }
    return ORBHIP_OK;
}

void* orbhip_matcher_stream(orbhip_matcher* m) { return m ? (void*)m->stream : nullptr; }
float orbhip_matcher_last_elapsed_ms(orbhip_matcher* m) { return m ? m->last_ms : 0.f; }

// Launch plan of the scan: query blocks, index-contiguous train chunks (enough workgroups to fill the
// chip several times over; a chunk never exceeds the key's index field) and the bytes of partial
// results the chunks exchange (0 with a single chunk).
// Default scan kernel of a new handle: the matrix-core one (knn2_mfma.h) unless ORBHIP_KNN_KERNEL=popcount.
static int default_scan_kernel() {
    const char* e = getenv("ORBHIP_KNN_KERNEL");
    return e && strcmp(e, "popcount") == 0 ? ORBHIP_SCAN_POPCOUNT : ORBHIP_SCAN_MATRIX;
}

int orbhip_matcher_set_scan_kernel(orbhip_matcher* m, int kind) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (kind != ORBHIP_SCAN_MATRIX && kind != ORBHIP_SCAN_POPCOUNT) return orb_fail(ORBHIP_EINVAL, "unknown scan kernel %d", kind);
    m->scan_kernel = kind;
    return ORBHIP_OK;
}

struct KnnPlan {
    int qblocks, nchunks, chunk_rows;
    size_t part_bytes;
};

static int knn2_check_and_plan(orbhip_matcher* m, const uint8_t* d_q, int nq, ptrdiff_t q_pair_stride, const uint8_t* d_t,
                               int nt, ptrdiff_t t_pair_stride, int npairs, const int32_t* d_idx, const int32_t* d_d1,
                               const int32_t* d_d2, KnnPlan* plan) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (nq < 0 || nt < 0 || npairs < 0) return orb_fail(ORBHIP_EINVAL, "negative size");
    plan->qblocks = plan->nchunks = plan->chunk_rows = 0;
    plan->part_bytes = 0;
    if (nq == 0 || npairs == 0) return ORBHIP_OK;
    if (!d_q || !d_idx || !d_d1 || !d_d2 || (nt > 0 && !d_t)) return orb_fail(ORBHIP_EINVAL, "null buffer");
    if (((uintptr_t)d_q & 15) || ((uintptr_t)d_t & 3) || (q_pair_stride & 15) || (t_pair_stride & 3))
        return orb_fail(ORBHIP_EINVAL, "descriptor buffers must be 16-byte (queries) / 4-byte (train) aligned");
    if (npairs > 65535) return orb_fail(ORBHIP_EINVAL, "more than 65535 pairs in one call");
    const int qblocks = (nq + 255) / 256;
    // Few (query block, pair) units: cut the train set into chunks so that the grid fills the chip KNN_BLOCKS_PER_CU
    // workgroups deep (every chunk the same number of rows).  A chunk keeps >= 2048 rows, so the unconditional
    // folds of its first rows stay a small share.  Many units: one chunk each.
    const long long units = (long long)qblocks * npairs;
    // (the matrix-core kernel holds four workgroups per CU and loses least with exactly one round of them: its
    // folds grow rarer the longer a chunk is)
    static const int per_cu_env = [] {
        const char* e = getenv("ORBHIP_KNN_BLOCKS_PER_CU");  // tuning experiments only
        return e ? atoi(e) : 0;
    }();
    const int per_cu = per_cu_env > 0 ? per_cu_env : (m->scan_kernel == ORBHIP_SCAN_MATRIX ? 4 : KNN_BLOCKS_PER_CU);
    const long long resident = (long long)per_cu * m->sm_count;
    int nchunks = (int)std::max<long long>(1, std::min<long long>(resident / units, (nt + 2047) / 2048));
    // a chunk is a whole number of 8-row groups: its first row is then 256-byte aligned relative to the set (an
    // odd row count would start every other chunk in the middle of a 64-byte line, and the scan's 64-byte scalar
    // loads would each straddle two lines: measured 2.30 against 2.13 ms on 4096 x 1.25 M)
    // (32 rows: a whole number of the matrix-core kernel's tiles as well)
    const long long cq = 32;
    int chunk_rows = nt > 0 ? (int)((((long long)nt + nchunks - 1) / nchunks + cq - 1) & ~(cq - 1)) : (int)cq;
    if (chunk_rows > (int)(KNN_IDX_MASK & ~(unsigned)(cq - 1))) chunk_rows = (int)(KNN_IDX_MASK & ~(unsigned)(cq - 1));
    nchunks = nt > 0 ? (nt + chunk_rows - 1) / chunk_rows : 1;
    if (nchunks > 65535) return orb_fail(ORBHIP_EINVAL, "train set too large");
    plan->qblocks = qblocks;
    plan->nchunks = nchunks;
    plan->chunk_rows = chunk_rows;
    plan->part_bytes = nchunks > 1 ? (size_t)npairs * nchunks * nq * sizeof(uint2) : 0;
    return ORBHIP_OK;
}

static int knn2_launch(orbhip_matcher* m, const KnnPlan& plan, const uint8_t* d_q, int nq, ptrdiff_t q_pair_stride,
                       const uint8_t* d_t, int nt, ptrdiff_t t_pair_stride, int npairs, int idx_base, uint2* d_part,
                       int32_t* d_idx, int32_t* d_d1, int32_t* d_d2) {
    HIPCHK(hipEventRecord(m->ev0, m->stream));
    if (m->scan_kernel == ORBHIP_SCAN_MATRIX)
        orb_launch_knn2_matrix(m->stream, plan.qblocks, plan.nchunks, npairs, (const uint32_t*)d_q, nq,
                               (long long)(q_pair_stride / 4), (const uint32_t*)d_t, nt, (long long)(t_pair_stride / 4),
                               plan.chunk_rows, idx_base, d_part, d_idx, d_d1, d_d2);
    else
        hipLaunchKernelGGL(k_knn2, dim3(plan.qblocks, plan.nchunks, npairs), dim3(256), 0, m->stream, (const uint32_t*)d_q, nq,
                           (long long)(q_pair_stride / 4), (const uint32_t*)d_t, nt, (long long)(t_pair_stride / 4),
                           plan.chunk_rows, plan.nchunks, idx_base, d_part, d_idx, d_d1, d_d2);
    if (plan.nchunks > 1)
        hipLaunchKernelGGL(k_knn2_fold, dim3(plan.qblocks, npairs), dim3(256), 0, m->stream, (const uint2*)d_part, nq,
                           plan.nchunks, plan.chunk_rows, idx_base, d_idx, d_d1, d_d2);
    HIPCHK(hipEventRecord(m->ev1, m->stream));
    HIPCHK(hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_knn2_device(orbhip_matcher* m, const uint8_t* d_q, int nq, ptrdiff_t q_pair_stride, const uint8_t* d_t, int nt,
                       ptrdiff_t t_pair_stride, int npairs, int idx_base, int32_t* d_idx, int32_t* d_d1, int32_t* d_d2) {
    KnnPlan plan;
    const int rc = knn2_check_and_plan(m, d_q, nq, q_pair_stride, d_t, nt, t_pair_stride, npairs, d_idx, d_d1, d_d2, &plan);
    if (rc || plan.qblocks == 0) return rc;
    HIPCHK(hipSetDevice(m->device));
    if (plan.part_bytes > m->part_bytes) {
        HIPCHK(hipStreamSynchronize(m->stream));
        if (m->d_part) (void)hipFree(m->d_part);
        m->d_part = nullptr;
        m->part_bytes = 0;
        if (hipMalloc(&m->d_part, plan.part_bytes) != hipSuccess)
            return orb_fail(ORBHIP_ENOMEM, "hipMalloc of %zu bytes failed", plan.part_bytes);
        m->part_bytes = plan.part_bytes;
    }
    return knn2_launch(m, plan, d_q, nq, q_pair_stride, d_t, nt, t_pair_stride, npairs, idx_base, (uint2*)m->d_part, d_idx,
                       d_d1, d_d2);
}

int orbhip_knn2_scratch_bytes(void) { return (int)KNN_SCRATCH_BYTES; }

int orbhip_knn2_device_noalloc(orbhip_matcher* m, const uint8_t* d_q, int nq, ptrdiff_t q_pair_stride, const uint8_t* d_t,
                               int nt, ptrdiff_t t_pair_stride, int npairs, int idx_base, int32_t* d_idx, int32_t* d_d1,
                               int32_t* d_d2) {
    KnnPlan plan;
    const int rc = knn2_check_and_plan(m, d_q, nq, q_pair_stride, d_t, nt, t_pair_stride, npairs, d_idx, d_d1, d_d2, &plan);
    if (rc || plan.qblocks == 0) return rc;
    if (!m->d_scratch) return orb_fail(ORBHIP_EHIP, "the device's scratch array was not resolved at matcher creation");
    if (plan.part_bytes > KNN_SCRATCH_BYTES)
        return orb_fail(ORBHIP_ECAP,
                        "partial results of this call need %zu bytes, the fixed scratch array holds %u "
                        "(npairs * chunks * nq * 8 bytes; use orbhip_knn2_device or fewer queries per call)",
                        plan.part_bytes, (unsigned)KNN_SCRATCH_BYTES);
    HIPCHK(hipSetDevice(m->device));
    if (plan.nchunks == 1)  // a single chunk writes its results directly: the array is not touched
        return knn2_launch(m, plan, d_q, nq, q_pair_stride, d_t, nt, t_pair_stride, npairs, idx_base, (uint2*)m->d_scratch, d_idx,
                           d_d1, d_d2);
    // Several handles (the reference's three threads each build their own matchers) share the device's one array:
    // this call's kernels wait for the previous user's, on whatever stream that was, and the next user waits for
    // these.  Neither call allocates.  Under stream capture the events would tie the captured graph to work outside
    // it, so a capturing caller gets no cross-handle ordering (replays are ordered by whoever launches the graphs).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(m->stream, &cap) != hipSuccess) {
        (void)hipGetLastError();
        cap = hipStreamCaptureStatusNone;
    }
    if (cap != hipStreamCaptureStatusNone)
        return knn2_launch(m, plan, d_q, nq, q_pair_stride, d_t, nt, t_pair_stride, npairs, idx_base, (uint2*)m->d_scratch, d_idx,
                           d_d1, d_d2);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);  // wait -> launch -> record is one step as seen by other host threads
    hipEvent_t ev = g_scratch_event[m->device];
    HIPCHK(hipStreamWaitEvent(m->stream, ev, 0));
    const int lrc = knn2_launch(m, plan, d_q, nq, q_pair_stride, d_t, nt, t_pair_stride, npairs, idx_base, (uint2*)m->d_scratch,
                                d_idx, d_d1, d_d2);
    HIPCHK(hipEventRecord(ev, m->stream));
    return lrc;
}

long long orbhip_knn2_partial_bytes(orbhip_matcher* m, int nq, int nt, int npairs) {
    if (!m || nq < 0 || nt < 0 || npairs < 0 || npairs > 65535) return -1;
    if (nq == 0 || npairs == 0) return 0;
    // the plan depends on the sizes alone (and on the handle's scan kernel); the pointers are only checked for alignment
    KnnPlan plan;
    static const int32_t dummy[4] = {0, 0, 0, 0};
    alignas(16) static const uint8_t aligned[16] = {0};
    if (knn2_check_and_plan(m, aligned, nq, 0, aligned, nt, 0, npairs, dummy, dummy, dummy, &plan)) return -1;
    return (long long)plan.part_bytes;
}

int orbhip_knn2(orbhip_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt, int npairs, int32_t* idx, int32_t* d1,
                int32_t* d2) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (nq < 0 || nt < 0 || npairs < 0) return orb_fail(ORBHIP_EINVAL, "negative size");
    if (nq == 0 || npairs == 0) return ORBHIP_OK;
    if (!q || !idx || !d1 || !d2 || (nt > 0 && !t)) return orb_fail(ORBHIP_EINVAL, "null buffer");
    HIPCHK(hipSetDevice(m->device));
    const size_t qb = (size_t)npairs * nq * 32, tb = (size_t)npairs * nt * 32, ob = (size_t)npairs * nq * 4;
    const size_t qb_al = (qb + 255) & ~(size_t)255, tb_al = (tb + 255) & ~(size_t)255, ob_al = (ob + 255) & ~(size_t)255;
    const size_t need = qb_al + tb_al + 3 * ob_al + 256;
    if (need > m->stage_bytes) {
        HIPCHK(hipStreamSynchronize(m->stream));
        if (m->d_stage) (void)hipFree(m->d_stage);
        m->d_stage = nullptr;
        m->stage_bytes = 0;
        if (hipMalloc(&m->d_stage, need) != hipSuccess) return orb_fail(ORBHIP_ENOMEM, "hipMalloc of %zu bytes failed", need);
        m->stage_bytes = need;
    }
    uint8_t* base = (uint8_t*)m->d_stage;
    uint8_t* dq = base;
    uint8_t* dt = base + qb_al;
    int32_t* di = (int32_t*)(dt + tb_al);
    int32_t* dd1 = (int32_t*)((uint8_t*)di + ob_al);
    int32_t* dd2 = (int32_t*)((uint8_t*)dd1 + ob_al);
    HIPCHK(hipMemcpyAsync(dq, q, qb, hipMemcpyHostToDevice, m->stream));
    if (tb) HIPCHK(hipMemcpyAsync(dt, t, tb, hipMemcpyHostToDevice, m->stream));
    int rc = orbhip_knn2_device(m, dq, nq, (ptrdiff_t)nq * 32, dt, nt, (ptrdiff_t)nt * 32, npairs, 0, di, dd1, dd2);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(idx, di, ob, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipMemcpyAsync(d1, dd1, ob, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipMemcpyAsync(d2, dd2, ob, hipMemcpyDeviceToHost, m->stream));
    return orbhip_matcher_synchronize(m);
}

int orbhip_knn2_merge_device(orbhip_matcher* m, const int32_t* d_parts, int nshards, int nq, int32_t* d_idx, int32_t* d_d1,
                             int32_t* d_d2) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (nshards < 1 || nq < 0) return orb_fail(ORBHIP_EINVAL, "bad size");
    if (nq == 0) return ORBHIP_OK;
    if (!d_parts || !d_idx || !d_d1 || !d_d2) return orb_fail(ORBHIP_EINVAL, "null buffer");
    HIPCHK(hipSetDevice(m->device));
    hipLaunchKernelGGL(k_knn2_merge, dim3((nq + 255) / 256), dim3(256), 0, m->stream, d_parts, nshards, nq, d_idx, d_d1, d_d2);
    HIPCHK(hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_ratio_test_device(orbhip_matcher* m, const int32_t* d_d1, const int32_t* d_d2, int n, float ratio, int form, int th,
                             uint8_t* d_accept) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (n < 0 || (form != 0 && form != 1)) return orb_fail(ORBHIP_EINVAL, "bad argument");
    if (n == 0) return ORBHIP_OK;
    if (!d_d1 || !d_d2 || !d_accept) return orb_fail(ORBHIP_EINVAL, "null buffer");
    HIPCHK(hipSetDevice(m->device));
    hipLaunchKernelGGL(k_ratio, dim3((n + 255) / 256), dim3(256), 0, m->stream, d_d1, d_d2, n, ratio, form, th, d_accept);
    HIPCHK(hipGetLastError());
    return ORBHIP_OK;
}

// Sustained integer-VALU rate of this device in lane-operations per second (64 lanes x wave
// instructions), measured with HIP events: mode 0 v_bcnt_u32_b32, 1 the matcher's xor + bcnt mix,
// 2 v_add_u32.  Returns 0 on failure.
double orbhip_valu_rate(orbhip_matcher* m, int mode) {
    if (!m || mode < 0 || mode > 2) return 0.0;
    if (hipSetDevice(m->device) != hipSuccess) return 0.0;
    const int blocks = m->sm_count * 32, iters = 4096;
    uint32_t* d = nullptr;
    if (hipMalloc(&d, (size_t)blocks * 256 * 4) != hipSuccess) return 0.0;
    double best = 0.0;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(m->ev0, m->stream);
        hipLaunchKernelGGL(k_valu_rate, dim3(blocks), dim3(256), 0, m->stream, d, iters, mode, 12345u + rep);
        (void)hipEventRecord(m->ev1, m->stream);
        if (hipStreamSynchronize(m->stream) != hipSuccess) break;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, m->ev0, m->ev1) != hipSuccess || ms <= 0.f) break;
        const double ops = (double)blocks * 256 * iters * 8;
        best = std::max(best, ops / (ms * 1e-3));
    }
    (void)hipFree(d);
    return best;
}

}  // extern "C"
