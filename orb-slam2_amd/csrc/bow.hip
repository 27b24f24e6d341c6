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
