"""Multi-GPU partitioning of the two paths, one process per GPU.

Extraction and frame-pair matching shard by contiguous ranges of independent units (frames,
pairs) and need no data-path collective.  Frame-versus-map matching splits the map descriptors
into index-contiguous shards; every rank scans its shard for all queries and the per-shard top-2
records are exchanged with ONE all-gather (RCCL over xGMI when the backend is "nccl") and merged
with the reference's update rule in ascending shard order, which reproduces the serial ascending
scan of src/ORBmatcher.cc:461-485 exactly.  The payload is 12 bytes per query and rank, so the
collective is latency-bound, not link-bandwidth-bound.
"""


def shard_range(n, rank, world):
    """Contiguous [begin, end) of n units owned by `rank`: the first n % world ranks get one more."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_layout(nq, world, rank):
    """orbhip_knn2_gather_layout: (int32 per record, int32 offset of rank's record, bytes of the receive buffer) of
    the all-gather of orbhip_knn2_sharded."""
    import ctypes as C
    from . import _lib
    rec, off, total = C.c_size_t(), C.c_size_t(), C.c_size_t()
    _lib.check(_lib.lib().orbhip_knn2_gather_layout(nq, world, rank, C.byref(rec), C.byref(off), C.byref(total)))
    return rec.value, off.value, total.value


def merge_host(parts, out=None):
    """orbhip_knn2_merge_host: the ordered merge of orbhip_knn2_merge_device on records in host memory.
    parts: int32 numpy array [nshards, 3, nq] (idx, d1, d2 per index-contiguous shard, ascending shard order).
    Returns int32 [3, nq]."""
    import ctypes as C
    import numpy as np
    from . import _lib
    parts = np.ascontiguousarray(parts, dtype=np.int32)
    nshards, three, nq = parts.shape
    assert three == 3
    if out is None:
        out = np.empty((3, nq), np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    _lib.check(_lib.lib().orbhip_knn2_merge_host(vp(parts), nshards, nq, vp(out[0]), vp(out[1]), vp(out[2])))
    return out


class Communicator:
    """orbhip_comm: one RCCL communicator (ncclComm_t) created inside liborbhip.so, one rank per process
    and GPU.  unique_id: the 128 bytes of orbhip_comm_unique_id() of ONE rank, handed to all ranks."""

    def __init__(self, unique_id, rank, world, device):
        import ctypes as C
        from . import _lib
        self._L = _lib.lib()
        self._h = C.c_void_p()
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        _lib.check(self._L.orbhip_comm_create(C.byref(self._h), buf, 128, rank, world, device))
        self.rank, self.world, self.device = rank, world, device

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _lib
        buf = (C.c_ubyte * 128)()
        _lib.check(_lib.lib().orbhip_comm_unique_id(buf, 128))
        return bytes(buf)

    @classmethod
    def from_torch_distributed(cls, dist, device, group=None):
        """The id travels through the process group that already exists (any backend); the communicator
        itself is liborbhip.so's own."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(box[0], rank, world, device)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.orbhip_comm_destroy(self._h)
            self._h.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def knn2_sharded_rccl(matcher, comm, d_q, nq, d_t_shard, nt_shard, idx_base, out):
    """orbhip_knn2_sharded: scan of this rank's shard, one RCCL all-gather and the ordered merge, all
    queued on the matcher's stream by the library (no torch collective, no host synchronisation).
    comm: Communicator or None (single shard).  out: int32 CUDA tensor [3, nq] (idx, d1, d2), identical
    on every rank after matcher.synchronize()."""
    from . import _lib
    _lib.check(matcher._L.orbhip_knn2_sharded(matcher._h, None if comm is None else comm._h, d_q.data_ptr(), nq,
                                              d_t_shard.data_ptr(), nt_shard, idx_base, out[0].data_ptr(),
                                              out[1].data_ptr(), out[2].data_ptr()))
    return out


def knn2_sharded(matcher, d_q, nq, d_t_shard, nt_shard, idx_base, parts_local, parts_all, out, dist=None, group=None):
    """Frame-vs-map kNN on one rank.

    matcher       object with knn2_device / merge_device / synchronize (ORBmatcher)
    d_q           queries, replicated on every rank (torch uint8 tensor [nq, 32] on the device)
    d_t_shard     this rank's contiguous slice of the map descriptors ([nt_shard, 32])
    idx_base      global index of the slice's first row
    parts_local   int32 tensor [3, nq]  (idx, d1, d2 of this shard)
    parts_all     int32 tensor [world, 3, nq]
    out           int32 tensor [3, nq]  (merged idx, d1, d2; identical on every rank)
    """
    matcher.knn2_device(d_q.data_ptr(), nq, nq * 32, d_t_shard.data_ptr(), nt_shard, nt_shard * 32, 1, idx_base,
                        parts_local[0].data_ptr(), parts_local[1].data_ptr(), parts_local[2].data_ptr())
    matcher.synchronize()
    world = 1
    if dist is not None and dist.is_initialized():
        world = dist.get_world_size(group)
    if world > 1 and parts_local.is_cuda and dist.get_backend(group) == "gloo":
        # a transport that moves host memory only: the records go through the host and back (12 B per query and rank)
        mine = parts_local.cpu()
        gathered = mine.new_empty((world,) + tuple(mine.shape))
        dist.all_gather_into_tensor(gathered.view(-1), mine.view(-1), group=group)
        parts_all.copy_(gathered)
        _sync_current_stream(parts_all)
    elif world > 1:
        dist.all_gather_into_tensor(parts_all.view(-1), parts_local.view(-1), group=group)
        _sync_current_stream(parts_all)
    else:
        parts_all[0].copy_(parts_local)
        _sync_current_stream(parts_all)
    matcher.merge_device(parts_all.data_ptr(), world, nq, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
    matcher.synchronize()
    return out


def _sync_current_stream(t):
    if t.is_cuda:
        import torch
        torch.cuda.current_stream(t.device).synchronize()
