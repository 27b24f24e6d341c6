// Bag-of-words transform of ORB descriptors through the vocabulary tree, for gfx950:
//   TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)
//   (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1218-1259) with FORB::distance
//   (Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101) as the metric,
// as Frame::ComputeBoW / KeyFrame::ComputeBoW run it for every descriptor of a frame
// (src/Frame.cc:281-288, src/KeyFrame.cc:82-91).
//
// Sixteen lanes descend the tree for one feature: at every level each lane takes one child of the
// current node (the reference's vocabulary has 10 per node), computes the Hamming distance to the
// child's descriptor, and the group keeps the FIRST child attaining the minimum — the reference
// updates only on "d < best_d" while walking the children in order.  Children of a node have
// consecutive ids in a tree built or loaded level by level, so the 16 lanes read one contiguous
// 512-byte run of node descriptors.
#include <hip/hip_runtime.h>

#include <climits>
#include <vector>

#include "matcher_internal.h"

struct orbhip_vocabulary {
    int device = 0;
    int nnodes = 0, L = 0, nwords = 0;
    int32_t* d_child_start = nullptr;  // nnodes + 1
    int32_t* d_child_ids = nullptr;    // nnodes - 1
    uint8_t* d_desc = nullptr;         // nnodes * 32
    int32_t* d_word = nullptr;         // nnodes, -1 for inner nodes
    double* d_weight = nullptr;        // nnodes
};

namespace {

#define BOW_LPF 16  // lanes per feature

__global__ __launch_bounds__(256) void k_bow_transform(const int32_t* __restrict__ child_start,
                                                       const int32_t* __restrict__ child_ids,
                                                       const uint8_t* __restrict__ node_desc,
                                                       const int32_t* __restrict__ node_word,
                                                       const double* __restrict__ node_weight, int L, int levelsup,
                                                       const uint8_t* __restrict__ feat, int n, int32_t* __restrict__ word_out,
                                                       int32_t* __restrict__ node_out, double* __restrict__ weight_out) {
    const int i = (blockIdx.x * 256 + threadIdx.x) / BOW_LPF;
    const int li = threadIdx.x % BOW_LPF;
    const bool valid = i < n;
    uint4 f0 = make_uint4(0, 0, 0, 0), f1 = f0;
    if (valid) {
        const uint4* fp = (const uint4*)(feat + (long long)i * 32);
        f0 = fp[0];
        f1 = fp[1];
    }
    const int nid_level = L - levelsup;
    int nid = 0;       // "if(nid_level <= 0 && nid != NULL) *nid = 0; // root"
    int final_id = 0;  // root
    int current_level = 0;
    bool active = valid;
    while (__ballot(active)) {
        ++current_level;
        int cs = 0, ce = 0;
        if (active) {
            cs = child_start[final_id];
            ce = child_start[final_id + 1];
        }
        // key: distance above, position among the children below -> minimum = first child at the least distance
        unsigned long long best = 0xffffffffffffffffull;
        for (int c0 = cs; __ballot(c0 < ce); c0 += BOW_LPF) {
            const int c = c0 + li;
            if (c < ce) {
                const int id = child_ids[c];
                const uint4* dp = (const uint4*)(node_desc + (long long)id * 32);
                const uint4 a = dp[0], b = dp[1];
                const int d = __popc(a.x ^ f0.x) + __popc(a.y ^ f0.y) + __popc(a.z ^ f0.z) + __popc(a.w ^ f0.w) +
                              __popc(b.x ^ f1.x) + __popc(b.y ^ f1.y) + __popc(b.z ^ f1.z) + __popc(b.w ^ f1.w);
                const unsigned long long key = ((unsigned long long)d << 32) | (unsigned)(c - cs);
                best = min(best, key);
            }
        }
#pragma unroll
        for (int o = BOW_LPF / 2; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o));
        if (active) {
            final_id = child_ids[cs + (int)(best & 0xffffffffu)];
            if (current_level == nid_level) nid = final_id;
            active = child_start[final_id] != child_start[final_id + 1];  // !isLeaf()
        }
    }
    if (valid && li == 0) {
        word_out[i] = node_word[final_id];
        node_out[i] = nid;
        weight_out[i] = node_weight[final_id];
    }
}

template <class T>
int upload_new(T** dst, const T* src, size_t count) {
    if (hipMalloc((void**)dst, sizeof(T) * std::max<size_t>(count, 1)) != hipSuccess)
        return orb_fail(ORBHIP_ENOMEM, "hipMalloc of %zu bytes failed", sizeof(T) * count);
    if (count && hipMemcpy(*dst, src, sizeof(T) * count, hipMemcpyHostToDevice) != hipSuccess)
        return orb_fail(ORBHIP_EHIP, "vocabulary upload failed");
    return ORBHIP_OK;
}

}  // namespace

extern "C" {

int orbhip_vocabulary_create(orbhip_vocabulary** out, int device, int nnodes, const int32_t* parent, const uint8_t* desc,
                             const double* weight, int L) {
    if (!out) return orb_fail(ORBHIP_EINVAL, "null out pointer");
    *out = nullptr;
    if (nnodes < 2 || !parent || !desc || !weight || L < 1) return orb_fail(ORBHIP_EINVAL, "bad vocabulary");
    for (int i = 1; i < nnodes; i++)
        if (parent[i] < 0 || parent[i] >= i) return orb_fail(ORBHIP_EINVAL, "node %d: parent %d does not precede it", i, parent[i]);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return orb_fail(ORBHIP_ENODEV, "no HIP device");
    if (device < 0 || device >= ndev) return orb_fail(ORBHIP_EINVAL, "device %d of %d", device, ndev);
    HIPCHK(hipSetDevice(device));
    // children attach to their parent in node-id order (TemplatedVocabulary.h loadFromTextFile / HKmeansStep)
    std::vector<int32_t> start(nnodes + 1, 0), ids(nnodes - 1), word(nnodes, -1);
    for (int i = 1; i < nnodes; i++) start[parent[i] + 1]++;
    for (int i = 0; i < nnodes; i++) start[i + 1] += start[i];
    std::vector<int32_t> fill(start.begin(), start.end() - 1);
    for (int i = 1; i < nnodes; i++) ids[fill[parent[i]]++] = i;
    int wid = 0;
    for (int i = 1; i < nnodes; i++)
        if (start[i] == start[i + 1]) word[i] = wid++;  // leaves are the words, numbered in node order
    orbhip_vocabulary* v = new orbhip_vocabulary;
    v->device = device;
    v->nnodes = nnodes;
    v->L = L;
    v->nwords = wid;
    int rc;
    if ((rc = upload_new(&v->d_child_start, start.data(), start.size())) || (rc = upload_new(&v->d_child_ids, ids.data(), ids.size())) ||
        (rc = upload_new(&v->d_desc, desc, (size_t)nnodes * 32)) || (rc = upload_new(&v->d_word, word.data(), word.size())) ||
        (rc = upload_new(&v->d_weight, weight, (size_t)nnodes))) {
        orbhip_vocabulary_destroy(v);
        return rc;
    }
    *out = v;
    return ORBHIP_OK;
}

void orbhip_vocabulary_destroy(orbhip_vocabulary* v) {
    if (!v) return;
    hipSetDevice(v->device);
    if (v->d_child_start) hipFree(v->d_child_start);
    if (v->d_child_ids) hipFree(v->d_child_ids);
    if (v->d_desc) hipFree(v->d_desc);
    if (v->d_word) hipFree(v->d_word);
    if (v->d_weight) hipFree(v->d_weight);
    delete v;
}

int orbhip_vocabulary_words(const orbhip_vocabulary* v) { return v ? v->nwords : ORBHIP_EINVAL; }

int orbhip_bow_transform_device(orbhip_matcher* m, const orbhip_vocabulary* v, const uint8_t* d_desc, int n, int levelsup,
                                int32_t* d_word, int32_t* d_node, double* d_weight) {
    if (!m || !v) return orb_fail(ORBHIP_EINVAL, "null handle");
    if (n < 0) return orb_fail(ORBHIP_EINVAL, "negative size");
    if (n == 0) return ORBHIP_OK;
    if (!d_desc || !d_word || !d_node || !d_weight) return orb_fail(ORBHIP_EINVAL, "null buffer");
    if (m->device != v->device) return orb_fail(ORBHIP_EINVAL, "matcher on device %d, vocabulary on %d", m->device, v->device);
    HIPCHK(hipSetDevice(m->device));
    (void)hipGetLastError();
    const long long threads = (long long)n * BOW_LPF;
    hipLaunchKernelGGL(k_bow_transform, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, m->stream, v->d_child_start,
                       v->d_child_ids, v->d_desc, v->d_word, v->d_weight, v->L, levelsup, d_desc, n, d_word, d_node, d_weight);
    HIPCHK(hipGetLastError());
    return ORBHIP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// FeatureVector of a frame (Thirdparty/DBoW2/DBoW2/FeatureVector.cpp:31-45 driven by
// TemplatedVocabulary.h:1150-1163): features with weight > 0 grouped by node id, nodes ascending,
// feature indices ascending inside a node.  One workgroup per frame sorts 64-bit keys
// (node << 32 | feature) in LDS and cuts the sorted run at node changes.
namespace {

#define FV_THREADS 256

__global__ __launch_bounds__(FV_THREADS) void k_feature_vector(const int32_t* __restrict__ node, const double* __restrict__ weight,
                                                               const int32_t* __restrict__ n_arr, int cap, int node_cap,
                                                               int32_t* __restrict__ fv_nodes, int32_t* __restrict__ fv_start,
                                                               int32_t* __restrict__ fv_feat, int32_t* __restrict__ fv_count,
                                                               int* __restrict__ err) {
    extern __shared__ unsigned long long fkeys[];
    __shared__ int nvalid_sh, nnodes_sh;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n = min(max(n_arr[f], 0), cap);
    int np2 = 64;
    while (np2 < n) np2 <<= 1;
    if (tid == 0) {
        nvalid_sh = 0;
        nnodes_sh = 0;
    }
    for (int i = tid; i < np2; i += FV_THREADS) {
        unsigned long long key = 0xffffffffffffffffull;
        if (i < n && weight[(long long)f * cap + i] > 0) key = ((unsigned long long)(uint32_t)node[(long long)f * cap + i] << 32) | (uint32_t)i;
        fkeys[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (np2 >> 1); t += FV_THREADS) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int hi = lo | j;
                const unsigned long long a = fkeys[lo], b = fkeys[hi];
                const bool up = (lo & k) == 0;
                if ((a > b) == up) {
                    fkeys[lo] = b;
                    fkeys[hi] = a;
                }
            }
            __syncthreads();
        }
    // feature indices in sorted order; a node starts where the upper word changes.  Node slots are
    // claimed in sorted order by counting the starts before each position (block-wide scan).
    int32_t* feat = fv_feat + (long long)f * cap;
    int32_t* nodes = fv_nodes + (long long)f * node_cap;
    int32_t* start = fv_start + (long long)f * (node_cap + 1);
    __shared__ int wsum[FV_THREADS / 64];
    __shared__ int carry;
    if (tid == 0) carry = 0;
    __syncthreads();
    int nvalid = 0;
    for (int base = 0; base < np2; base += FV_THREADS) {
        const int i = base + tid;
        const unsigned long long key = i < np2 ? fkeys[i] : 0xffffffffffffffffull;
        const bool valid = key != 0xffffffffffffffffull;
        const bool head = valid && (i == 0 || (uint32_t)(fkeys[i - 1] >> 32) != (uint32_t)(key >> 32));
        if (valid) feat[i] = (int32_t)(key & 0xffffffffu);
        int incl = head ? 1 : 0;
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int off = carry;
        for (int w = 0; w < wave; w++) off += wsum[w];
        if (head) {
            const int slot = off + incl - 1;
            if (slot < node_cap) {
                nodes[slot] = (int32_t)(key >> 32);
                start[slot] = i;
            }
        }
        nvalid += valid ? 1 : 0;
        __syncthreads();
        if (tid == FV_THREADS - 1) carry = off + incl;
        __syncthreads();
    }
    // total number of valid features = position of the first invalid key (keys are sorted)
    atomicAdd(&nvalid_sh, nvalid);
    __syncthreads();
    if (tid == 0) {
        const int nn = carry;
        if (nn > node_cap) atomicOr(err, 16);
        const int kept = min(nn, node_cap);
        start[kept] = nvalid_sh;
        fv_count[f] = kept;
    }
}

// ---------------------------------------------------------------------------------------------
// SearchByBoW (src/ORBmatcher.cc:158-295 and :726-861): features under the same vocabulary node are
// matched with the best / second-best scan; a feature of the second frame is taken at most once.
// Nodes are independent (a feature lies under exactly one node), so one wave handles one node of
// the first frame: it finds the node in the second frame's sorted node list, walks the first
// frame's features of the node in order and lets its 64 lanes examine the second frame's list.
#define BOW_HISTO_LENGTH 30
#define BOW_TH_LOW 20

struct KpDevB {
    float x, y, size, angle, response;
    int32_t octave, class_id;
};

__global__ __launch_bounds__(64) void k_bow_match(const uint8_t* __restrict__ desc1, const uint8_t* __restrict__ valid1,
                                                  const int32_t* __restrict__ fv1_nodes, const int32_t* __restrict__ fv1_start,
                                                  const int32_t* __restrict__ fv1_feat, const int32_t* __restrict__ fv1_count,
                                                  const uint8_t* __restrict__ desc2, const uint8_t* __restrict__ valid2,
                                                  const int32_t* __restrict__ fv2_nodes, const int32_t* __restrict__ fv2_start,
                                                  const int32_t* __restrict__ fv2_feat, const int32_t* __restrict__ fv2_count,
                                                  int cap, int node_cap, float ratio, int th_inclusive,
                                                  int32_t* __restrict__ matches12, int32_t* __restrict__ matches21,
                                                  int32_t* __restrict__ nmatches) {
    __shared__ uint8_t taken[ORBHIP_FRAME_MAX_KEYPOINTS];
    const int j = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
    if (j >= fv1_count[f]) return;
    const int node = fv1_nodes[(long long)f * node_cap + j];
    // lower bound of `node` in the second frame's node list
    const int32_t* n2 = fv2_nodes + (long long)f * node_cap;
    int lo = 0, hi = fv2_count[f];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (n2[mid] < node) lo = mid + 1;
        else hi = mid;
    }
    if (lo >= fv2_count[f] || n2[lo] != node) return;
    const int32_t* s1 = fv1_start + (long long)f * (node_cap + 1);
    const int32_t* s2 = fv2_start + (long long)f * (node_cap + 1);
    const int32_t* l1 = fv1_feat + (long long)f * cap + s1[j];
    const int32_t* l2 = fv2_feat + (long long)f * cap + s2[lo];
    const int len1 = s1[j + 1] - s1[j], len2 = s2[lo + 1] - s2[lo];
    for (int p = lane; p < len2; p += 64) taken[p] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint8_t* v1 = valid1 + (long long)f * cap;
    const uint8_t* v2 = valid2 ? valid2 + (long long)f * cap : nullptr;
    const uint8_t* d1f = desc1 + (long long)f * cap * 32;
    const uint8_t* d2f = desc2 + (long long)f * cap * 32;
    int32_t* m12 = matches12 + (long long)f * cap;
    int32_t* m21 = matches21 + (long long)f * cap;
    int count = 0;
    for (int i1 = 0; i1 < len1; i1++) {
        const int idx1 = l1[i1];
        if (!v1[idx1]) continue;  // no map point / bad map point
        const uint4* q = (const uint4*)(d1f + (long long)idx1 * 32);
        const uint4 q0 = q[0], q1 = q[1];
        unsigned long long k1 = 0xffffffffffffffffull, k2 = 0xffffffffffffffffull;
        for (int p = lane; p < len2; p += 64) {
            const int idx2 = l2[p];
            if (taken[p] || (v2 && !v2[idx2])) continue;
            const uint4* t = (const uint4*)(d2f + (long long)idx2 * 32);
            const uint4 a = t[0], b = t[1];
            const int d = __popc(a.x ^ q0.x) + __popc(a.y ^ q0.y) + __popc(a.z ^ q0.z) + __popc(a.w ^ q0.w) +
                          __popc(b.x ^ q1.x) + __popc(b.y ^ q1.y) + __popc(b.z ^ q1.z) + __popc(b.w ^ q1.w);
            const unsigned long long key = ((unsigned long long)d << 32) | (unsigned)p;
            k2 = min(max(k1, key), k2);
            k1 = min(k1, key);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long p1 = __shfl_xor(k1, o), p2 = __shfl_xor(k2, o);
            const unsigned long long mn = min(k1, p1), mx = max(k1, p1);
            k2 = min(mx, min(k2, p2));
            k1 = mn;
        }
        if (k1 == 0xffffffffffffffffull) continue;
        const int bestDist1 = (int)(k1 >> 32), pos = (int)(k1 & 0xffffffffu);
        const float fBest2 = k2 == 0xffffffffffffffffull ? (float)INT_MAX : (float)(int)(k2 >> 32);
        const bool low = th_inclusive ? bestDist1 <= BOW_TH_LOW : bestDist1 < BOW_TH_LOW;  // :235 / :802
        if (low && (float)bestDist1 < __fmul_rn(ratio, fBest2)) {                            // :237 / :804
            if (lane == 0) {
                const int idx2 = l2[pos];
                m12[idx1] = idx2;
                m21[idx2] = idx1;
                taken[pos] = 1;
            }
            count++;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
    if (lane == 0 && count) atomicAdd(&nmatches[f], count);
}

// SearchForTriangulation (src/ORBmatcher.cc:863-1025): as in k_bow_match one wave per vocabulary node that both
// keyframes share.  For a feature of keyframe 1 without a map point the reference collects the features of the
// node in keyframe 2 that are free (no map point, not matched by an earlier feature) and at most TH_LOW away, sorts
// them by (distance, index) and takes the first one within 2 * the least distance that passes
// CheckDistEpipolarLine (:139-156).  Here every lane tests its candidates (distance, epipolar line) and the wave
// reduces two keys: the least (distance, index) overall and the least among those on the epipolar line.
__global__ __launch_bounds__(64) void k_bow_triangulate(const KpDevB* __restrict__ kp1, const uint8_t* __restrict__ desc1,
                                                        const uint8_t* __restrict__ free1, const int32_t* __restrict__ fv1_nodes,
                                                        const int32_t* __restrict__ fv1_start, const int32_t* __restrict__ fv1_feat,
                                                        const int32_t* __restrict__ fv1_count, const KpDevB* __restrict__ kp2,
                                                        const uint8_t* __restrict__ desc2, const uint8_t* __restrict__ free2,
                                                        const int32_t* __restrict__ fv2_nodes, const int32_t* __restrict__ fv2_start,
                                                        const int32_t* __restrict__ fv2_feat, const int32_t* __restrict__ fv2_count,
                                                        int cap, int node_cap, const float* __restrict__ F12,
                                                        const float* __restrict__ level_sigma2, int nlevels,
                                                        int32_t* __restrict__ matches12, int32_t* __restrict__ matches21,
                                                        int32_t* __restrict__ nmatches) {
    __shared__ uint8_t taken[ORBHIP_FRAME_MAX_KEYPOINTS];
    const int j = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
    if (j >= fv1_count[f]) return;
    const int node = fv1_nodes[(long long)f * node_cap + j];
    const int32_t* n2 = fv2_nodes + (long long)f * node_cap;
    int lo = 0, hi = fv2_count[f];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (n2[mid] < node) lo = mid + 1;
        else hi = mid;
    }
    if (lo >= fv2_count[f] || n2[lo] != node) return;
    const int32_t* s1 = fv1_start + (long long)f * (node_cap + 1);
    const int32_t* s2 = fv2_start + (long long)f * (node_cap + 1);
    const int32_t* l1 = fv1_feat + (long long)f * cap + s1[j];
    const int32_t* l2 = fv2_feat + (long long)f * cap + s2[lo];
    const int len1 = s1[j + 1] - s1[j], len2 = s2[lo + 1] - s2[lo];
    for (int p = lane; p < len2; p += 64) taken[p] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint8_t* f1 = free1 + (long long)f * cap;
    const uint8_t* f2 = free2 + (long long)f * cap;
    const uint8_t* d1f = desc1 + (long long)f * cap * 32;
    const uint8_t* d2f = desc2 + (long long)f * cap * 32;
    const KpDevB* k1f = kp1 + (long long)f * cap;
    const KpDevB* k2f = kp2 + (long long)f * cap;
    const float* F = F12 + (long long)f * 9;
    int32_t* m12 = matches12 + (long long)f * cap;
    int32_t* m21 = matches21 + (long long)f * cap;
    const unsigned long long NONE = 0xffffffffffffffffull;
    int count = 0;
    for (int i1 = 0; i1 < len1; i1++) {
        const int idx1 = l1[i1];
        if (!f1[idx1]) continue;  // "If there is already a MapPoint skip"
        const uint4* q = (const uint4*)(d1f + (long long)idx1 * 32);
        const uint4 q0 = q[0], q1 = q[1];
        // epipolar line in the second image, l = x1' F12 = [a b c] (:142-144), single precision, no contraction
        const float x1 = k1f[idx1].x, y1 = k1f[idx1].y;
        const float a = __fadd_rn(__fadd_rn(__fmul_rn(x1, F[0]), __fmul_rn(y1, F[3])), F[6]);
        const float b = __fadd_rn(__fadd_rn(__fmul_rn(x1, F[1]), __fmul_rn(y1, F[4])), F[7]);
        const float c = __fadd_rn(__fadd_rn(__fmul_rn(x1, F[2]), __fmul_rn(y1, F[5])), F[8]);
        const float den = __fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b));
        unsigned long long kA = NONE, kB = NONE;
        for (int p = lane; p < len2; p += 64) {
            const int idx2 = l2[p];
            if (taken[p] || !f2[idx2]) continue;  // "If we have already matched or there is a MapPoint skip"
            const uint4* t = (const uint4*)(d2f + (long long)idx2 * 32);
            const uint4 ta = t[0], tb = t[1];
            const int d = __popc(ta.x ^ q0.x) + __popc(ta.y ^ q0.y) + __popc(ta.z ^ q0.z) + __popc(ta.w ^ q0.w) +
                          __popc(tb.x ^ q1.x) + __popc(tb.y ^ q1.y) + __popc(tb.z ^ q1.z) + __popc(tb.w ^ q1.w);
            if (d > BOW_TH_LOW) continue;
            const unsigned long long key = ((unsigned long long)d << 32) | ((unsigned long long)idx2 << 16) | (unsigned)p;
            kA = key < kA ? key : kA;
            const KpDevB& k2 = k2f[idx2];
            const float num = __fadd_rn(__fadd_rn(__fmul_rn(a, k2.x), __fmul_rn(b, k2.y)), c);
            bool on_line = false;
            if (den != 0.f) {
                const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
                const int oct = min(max(k2.octave, 0), nlevels - 1);
                on_line = (double)dsqr < 3.84 * (double)level_sigma2[oct];
            }
            if (on_line) kB = key < kB ? key : kB;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long pA = __shfl_xor(kA, o), pB = __shfl_xor(kB, o);
            kA = pA < kA ? pA : kA;
            kB = pB < kB ? pB : kB;
        }
        if (kA == NONE || kB == NONE) continue;
        const int BestDist = (int)(kA >> 32);
        const int DistTh = 2 * BestDist;  // round(2*BestDist)
        if ((int)(kB >> 32) > DistTh) continue;
        if (lane == 0) {
            const int idx2 = (int)((kB >> 16) & 0xffffu), pos = (int)(kB & 0xffffu);
            m12[idx1] = idx2;
            m21[idx2] = idx1;
            taken[pos] = 1;
        }
        count++;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (lane == 0 && count) atomicAdd(&nmatches[f], count);
}

__device__ __forceinline__ float bow_round_half_away(float v) {
    const float t = truncf(v);
    return fabsf(__fsub_rn(v, t)) >= 0.5f ? (v < 0.f ? t - 1.f : t + 1.f) : t;
}
__device__ __forceinline__ int bow_rot_bin(float a1, float a2) {
    const float factor = 1.0f / BOW_HISTO_LENGTH;
    float rot = __fsub_rn(a1, a2);
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)bow_round_half_away(__fmul_rn(rot, factor));
    if (bin == BOW_HISTO_LENGTH) bin = 0;
    return bin;
}

// Rotation histogram over all matches, ComputeThreeMaxima (src/ORBmatcher.cc:1759-1800), removal of
// the matches outside the kept bins.  One workgroup per frame pair.
__global__ __launch_bounds__(256) void k_bow_rot_filter(const KpDevB* __restrict__ kp1, const KpDevB* __restrict__ kp2,
                                                        const int32_t* __restrict__ n1_arr, int cap,
                                                        int32_t* __restrict__ matches12, int32_t* __restrict__ matches21,
                                                        int32_t* __restrict__ nmatches) {
    __shared__ int hist[32];
    __shared__ int removed;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n1 = min(max(n1_arr[f], 0), cap);
    if (tid < 32) hist[tid] = 0;
    if (tid == 0) removed = 0;
    __syncthreads();
    const KpDevB* k1 = kp1 + (long long)f * cap;
    const KpDevB* k2 = kp2 + (long long)f * cap;
    int32_t* m12 = matches12 + (long long)f * cap;
    int32_t* m21 = matches21 + (long long)f * cap;
    for (int i = tid; i < n1; i += 256)
        if (m12[i] >= 0) atomicAdd(&hist[bow_rot_bin(k1[i].angle, k2[m12[i]].angle)], 1);
    __syncthreads();
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < BOW_HISTO_LENGTH; i++) {
        const int sz = hist[i];
        if (sz > max1) {
            max3 = max2; max2 = max1; max1 = sz;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (sz > max2) {
            max3 = max2; max2 = sz;
            ind3 = ind2; ind2 = i;
        } else if (sz > max3) {
            max3 = sz;
            ind3 = i;
        }
    }
    if ((float)max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
    } else if ((float)max3 < 0.1f * (float)max1) {
        ind3 = -1;
    }
    int mine = 0;
    for (int i = tid; i < n1; i += 256) {
        const int m = m12[i];
        if (m >= 0) {
            const int bin = bow_rot_bin(k1[i].angle, k2[m].angle);
            if (bin != ind1 && bin != ind2 && bin != ind3) {
                m12[i] = -1;
                m21[m] = -1;
                mine++;
            }
        }
    }
    if (mine) atomicAdd(&removed, mine);
    __syncthreads();
    if (tid == 0) nmatches[f] -= removed;
}

}  // namespace

extern "C" {

int orbhip_feature_vector_device(orbhip_matcher* m, const int32_t* d_node, const double* d_weight, const int32_t* d_n,
                                 int batch, int cap, int node_cap, int32_t* d_fv_nodes, int32_t* d_fv_start,
                                 int32_t* d_fv_feat, int32_t* d_fv_count) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (batch < 0 || cap < 1 || node_cap < 1) return orb_fail(ORBHIP_EINVAL, "bad size");
    if (cap > ORBHIP_FRAME_MAX_KEYPOINTS) return orb_fail(ORBHIP_ECAP, "cap %d > %d features per frame", cap, ORBHIP_FRAME_MAX_KEYPOINTS);
    if (batch == 0) return ORBHIP_OK;
    if (!d_node || !d_weight || !d_n || !d_fv_nodes || !d_fv_start || !d_fv_feat || !d_fv_count)
        return orb_fail(ORBHIP_EINVAL, "null buffer");
    HIPCHK(hipSetDevice(m->device));
    (void)hipGetLastError();
    int np2 = 64;
    while (np2 < cap) np2 <<= 1;
    const size_t lds = sizeof(unsigned long long) * (size_t)np2;
    HIPCHK(hipFuncSetAttribute((const void*)k_feature_vector, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_feature_vector, dim3(batch), dim3(FV_THREADS), lds, m->stream, d_node, d_weight, d_n, cap, node_cap,
                       d_fv_nodes, d_fv_start, d_fv_feat, d_fv_count, m->d_err);
    HIPCHK(hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_search_by_bow_device(orbhip_matcher* m, const OrbKeypoint* d_kp1, const uint8_t* d_desc1, const int32_t* d_n1,
                                const uint8_t* d_valid1, const int32_t* d_fv1_nodes, const int32_t* d_fv1_start,
                                const int32_t* d_fv1_feat, const int32_t* d_fv1_count, const OrbKeypoint* d_kp2,
                                const uint8_t* d_desc2, const int32_t* d_n2, const uint8_t* d_valid2,
                                const int32_t* d_fv2_nodes, const int32_t* d_fv2_start, const int32_t* d_fv2_feat,
                                const int32_t* d_fv2_count, int batch, int cap, int node_cap, float nnratio,
                                int check_orientation, int th_low_inclusive, int32_t* d_matches12, int32_t* d_matches21,
                                int32_t* d_nmatches) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (batch < 0 || cap < 1 || node_cap < 1) return orb_fail(ORBHIP_EINVAL, "bad size");
    if (cap > ORBHIP_FRAME_MAX_KEYPOINTS) return orb_fail(ORBHIP_ECAP, "cap %d > %d features per frame", cap, ORBHIP_FRAME_MAX_KEYPOINTS);
    if (batch == 0) return ORBHIP_OK;
    if (!d_kp1 || !d_desc1 || !d_n1 || !d_valid1 || !d_fv1_nodes || !d_fv1_start || !d_fv1_feat || !d_fv1_count || !d_kp2 ||
        !d_desc2 || !d_n2 || !d_fv2_nodes || !d_fv2_start || !d_fv2_feat || !d_fv2_count || !d_matches12 || !d_matches21 ||
        !d_nmatches)
        return orb_fail(ORBHIP_EINVAL, "null buffer");
    HIPCHK(hipSetDevice(m->device));
    (void)hipGetLastError();
    HIPCHK(hipEventRecord(m->ev0, m->stream));
    HIPCHK(hipMemsetAsync(d_matches12, 0xff, sizeof(int32_t) * (size_t)batch * cap, m->stream));
    HIPCHK(hipMemsetAsync(d_matches21, 0xff, sizeof(int32_t) * (size_t)batch * cap, m->stream));
    HIPCHK(hipMemsetAsync(d_nmatches, 0, sizeof(int32_t) * (size_t)batch, m->stream));
    hipLaunchKernelGGL(k_bow_match, dim3(node_cap, batch), dim3(64), 0, m->stream, d_desc1, d_valid1, d_fv1_nodes, d_fv1_start,
                       d_fv1_feat, d_fv1_count, d_desc2, d_valid2, d_fv2_nodes, d_fv2_start, d_fv2_feat, d_fv2_count, cap,
                       node_cap, nnratio, th_low_inclusive, d_matches12, d_matches21, d_nmatches);
    if (check_orientation)
        hipLaunchKernelGGL(k_bow_rot_filter, dim3(batch), dim3(256), 0, m->stream, (const KpDevB*)d_kp1, (const KpDevB*)d_kp2,
                           d_n1, cap, d_matches12, d_matches21, d_nmatches);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(m->ev1, m->stream));
    return ORBHIP_OK;
}

int orbhip_search_for_triangulation_device(orbhip_matcher* m, const OrbKeypoint* d_kp1, const uint8_t* d_desc1,
                                           const int32_t* d_n1, const uint8_t* d_free1, const int32_t* d_fv1_nodes,
                                           const int32_t* d_fv1_start, const int32_t* d_fv1_feat, const int32_t* d_fv1_count,
                                           const OrbKeypoint* d_kp2, const uint8_t* d_desc2, const int32_t* d_n2,
                                           const uint8_t* d_free2, const int32_t* d_fv2_nodes, const int32_t* d_fv2_start,
                                           const int32_t* d_fv2_feat, const int32_t* d_fv2_count, int batch, int cap, int node_cap,
                                           const float* d_F12, const float* d_level_sigma2, int nlevels, int check_orientation,
                                           int32_t* d_matches12, int32_t* d_matches21, int32_t* d_nmatches) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (batch < 0 || cap < 1 || node_cap < 1 || nlevels < 1) return orb_fail(ORBHIP_EINVAL, "bad size");
    if (cap > ORBHIP_FRAME_MAX_KEYPOINTS) return orb_fail(ORBHIP_ECAP, "cap %d > %d features per frame", cap, ORBHIP_FRAME_MAX_KEYPOINTS);
    if (batch == 0) return ORBHIP_OK;
    if (!d_kp1 || !d_desc1 || !d_n1 || !d_free1 || !d_fv1_nodes || !d_fv1_start || !d_fv1_feat || !d_fv1_count || !d_kp2 ||
        !d_desc2 || !d_n2 || !d_free2 || !d_fv2_nodes || !d_fv2_start || !d_fv2_feat || !d_fv2_count || !d_F12 || !d_level_sigma2 ||
        !d_matches12 || !d_matches21 || !d_nmatches)
        return orb_fail(ORBHIP_EINVAL, "null buffer");
    HIPCHK(hipSetDevice(m->device));
    (void)hipGetLastError();
    HIPCHK(hipEventRecord(m->ev0, m->stream));
    HIPCHK(hipMemsetAsync(d_matches12, 0xff, sizeof(int32_t) * (size_t)batch * cap, m->stream));
    HIPCHK(hipMemsetAsync(d_matches21, 0xff, sizeof(int32_t) * (size_t)batch * cap, m->stream));
    HIPCHK(hipMemsetAsync(d_nmatches, 0, sizeof(int32_t) * (size_t)batch, m->stream));
    hipLaunchKernelGGL(k_bow_triangulate, dim3(node_cap, batch), dim3(64), 0, m->stream, (const KpDevB*)d_kp1, d_desc1, d_free1,
                       d_fv1_nodes, d_fv1_start, d_fv1_feat, d_fv1_count, (const KpDevB*)d_kp2, d_desc2, d_free2, d_fv2_nodes,
                       d_fv2_start, d_fv2_feat, d_fv2_count, cap, node_cap, d_F12, d_level_sigma2, nlevels, d_matches12,
                       d_matches21, d_nmatches);
    if (check_orientation)
        hipLaunchKernelGGL(k_bow_rot_filter, dim3(batch), dim3(256), 0, m->stream, (const KpDevB*)d_kp1, (const KpDevB*)d_kp2,
                           d_n1, cap, d_matches12, d_matches21, d_nmatches);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(m->ev1, m->stream));
    return ORBHIP_OK;
}

}  // extern "C"
