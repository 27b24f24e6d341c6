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
    if world > 1:
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
