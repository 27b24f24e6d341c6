// Frame-versus-map Hamming kNN over a map that is split into index-contiguous shards, one shard per
// GPU and process: every rank scans its shard for all queries (the best / second-best scan of
// src/ORBmatcher.cc:461-485 with DescriptorDistance, :1805-1821, as the metric), the per-shard
// (idx, d1, d2) records are exchanged with ONE RCCL all-gather over xGMI, and every rank folds the
// shards in ascending rank order with the reference's update rule, which reproduces the serial
// ascending scan exactly.  The payload is 12 bytes per query and rank (48 KB for 4000 queries), so the
// collective is latency-bound; scan, all-gather and merge are queued on the matcher's stream without
// a host synchronisation in between.
//
// RCCL is bound at run time (dlopen) so that the library has no link-time dependency on it and
// shares whatever librccl the host process has already loaded (a second copy of RCCL in one process
// would keep its own topology state).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/orbhip.h"
#include "common.h"
#include "knn2_merge.h"
#include "matcher_internal.h"

namespace {

struct RcclApi {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

RcclApi g_rccl;
std::mutex g_rccl_mutex;

// The copy of RCCL the process already holds wins (RTLD_NOLOAD); otherwise ORBHIP_RCCL_LIB, then the
// system library.
int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.ok) return ORBHIP_OK;
    void* so = nullptr;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names)
        if (!so) so = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (!so) {
        const char* env = getenv("ORBHIP_RCCL_LIB");
        if (env && *env) so = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    }
    for (const char* n : names)
        if (!so) so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!so) return orb_fail(ORBHIP_EUNSUPPORTED, "librccl.so.1 could not be loaded: %s", dlerror());
    RcclApi a;
    a.so = so;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(so, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(so, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(so, "ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather))dlsym(so, "ncclAllGather");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(so, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.GetErrorString)
        return orb_fail(ORBHIP_EUNSUPPORTED, "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather");
    a.ok = true;
    g_rccl = a;
    return ORBHIP_OK;
}

#define RCCLCHK(call)                                                                                  \
    do {                                                                                               \
        ncclResult_t r__ = (call);                                                                     \
        if (r__ != ncclSuccess) return orb_fail(ORBHIP_EHIP, "%s failed: %s", #call, g_rccl.GetErrorString(r__)); \
    } while (0)

}  // namespace

struct orbhip_comm {
    ncclComm_t nccl = nullptr;
    int rank = 0, world = 1, device = 0;
    bool owned = false;
};

extern "C" {

int orbhip_comm_unique_id(void* id_out, size_t id_bytes) {
    if (!id_out || id_bytes < sizeof(ncclUniqueId))
        return orb_fail(ORBHIP_EINVAL, "the id buffer must hold %zu bytes", sizeof(ncclUniqueId));
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    RCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return ORBHIP_OK;
}

int orbhip_comm_create(orbhip_comm** out, const void* id, size_t id_bytes, int rank, int world, int device) {
    if (!out) return orb_fail(ORBHIP_EINVAL, "null out pointer");
    *out = nullptr;
    if (!id || id_bytes < sizeof(ncclUniqueId)) return orb_fail(ORBHIP_EINVAL, "the id must hold %zu bytes", sizeof(ncclUniqueId));
    if (world < 1 || rank < 0 || rank >= world) return orb_fail(ORBHIP_EINVAL, "rank %d of %d", rank, world);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return orb_fail(ORBHIP_ENODEV, "no HIP device");
    if (device < 0 || device >= ndev) return orb_fail(ORBHIP_EINVAL, "device %d of %d", device, ndev);
    int rc = load_rccl();
    if (rc) return rc;
    HIPCHK(hipSetDevice(device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    RCCLCHK(g_rccl.CommInitRank(&c, world, uid, rank));
    orbhip_comm* h = new orbhip_comm;
    h->nccl = c;
    h->rank = rank;
    h->world = world;
    h->device = device;
    h->owned = true;
    *out = h;
    return ORBHIP_OK;
}

int orbhip_comm_adopt(orbhip_comm** out, void* nccl_comm, int rank, int world, int device) {
    if (!out) return orb_fail(ORBHIP_EINVAL, "null out pointer");
    *out = nullptr;
    if (!nccl_comm) return orb_fail(ORBHIP_EINVAL, "null communicator");
    if (world < 1 || rank < 0 || rank >= world) return orb_fail(ORBHIP_EINVAL, "rank %d of %d", rank, world);
    int rc = load_rccl();
    if (rc) return rc;
    orbhip_comm* h = new orbhip_comm;
    h->nccl = (ncclComm_t)nccl_comm;
    h->rank = rank;
    h->world = world;
    h->device = device;
    h->owned = false;
    *out = h;
    return ORBHIP_OK;
}

void orbhip_comm_destroy(orbhip_comm* c) {
    if (!c) return;
    if (c->owned && c->nccl && g_rccl.ok) {
        (void)hipSetDevice(c->device);
        g_rccl.CommDestroy(c->nccl);
    }
    delete c;
}

int orbhip_comm_rank(const orbhip_comm* c) { return c ? c->rank : 0; }
int orbhip_comm_world(const orbhip_comm* c) { return c ? c->world : 1; }

int orbhip_knn2_sharded(orbhip_matcher* m, orbhip_comm* comm, const uint8_t* d_q, int nq, const uint8_t* d_t_shard,
                        int nt_shard, int idx_base, int32_t* d_idx, int32_t* d_d1, int32_t* d_d2) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (nq < 0 || nt_shard < 0 || idx_base < 0) return orb_fail(ORBHIP_EINVAL, "negative size");
    if (nq == 0) return ORBHIP_OK;
    if (!d_q || !d_idx || !d_d1 || !d_d2) return orb_fail(ORBHIP_EINVAL, "null buffer");
    const int world = comm ? comm->world : 1, rank = comm ? comm->rank : 0;
    if (comm && comm->device != m->device)
        return orb_fail(ORBHIP_EINVAL, "communicator on device %d, matcher on device %d", comm->device, m->device);
    if ((long long)idx_base + nt_shard > 2147483647LL) return orb_fail(ORBHIP_EINVAL, "global row index exceeds int32");
    HIPCHK(hipSetDevice(m->device));
    if (!comm) {
        // one shard and no communicator: the scan's result is the answer
        return orbhip_knn2_device(m, d_q, nq, 0, d_t_shard, nt_shard, 0, 1, idx_base, d_idx, d_d1, d_d2);
    }
    const size_t rec = knn2_record_ints(nq);  // int32 per rank: idx[nq], d1[nq], d2[nq]
    const size_t need = rec * world * sizeof(int32_t);
    if (need > m->gather_bytes) {
        HIPCHK(hipStreamSynchronize(m->stream));
        if (m->d_gather) (void)hipFree(m->d_gather);
        m->d_gather = nullptr;
        m->gather_bytes = 0;
        if (hipMalloc(&m->d_gather, need) != hipSuccess) return orb_fail(ORBHIP_ENOMEM, "hipMalloc of %zu bytes failed", need);
        m->gather_bytes = need;
    }
    int32_t* all = (int32_t*)m->d_gather;
    int32_t* mine = all + knn2_slot_offset_ints(nq, rank);
    int rc = orbhip_knn2_device(m, d_q, nq, 0, d_t_shard, nt_shard, 0, 1, idx_base, mine, mine + nq, mine + 2 * (size_t)nq);
    if (rc) return rc;
    // in place: this rank's record already sits at its slot of the receive buffer
    RCCLCHK(g_rccl.AllGather(mine, all, rec, ncclInt32, comm->nccl, m->stream));
    rc = orbhip_knn2_merge_device(m, all, world, nq, d_idx, d_d1, d_d2);
    if (rc) return rc;
    HIPCHK(hipEventRecord(m->ev1, m->stream));  // last_elapsed_ms covers scan + all-gather + merge
    return ORBHIP_OK;
}

}  // extern "C"
