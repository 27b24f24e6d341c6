"""Frame-versus-map matching over index-contiguous map shards, with the real HIP matcher on device
tensors: the torch.distributed form (sharded.knn2_sharded), the C entry point with its own RCCL
communicator (orbhip_knn2_sharded), and full-size shards (4000 queries x 1.25 M rows each) merged on
the device.  One GPU here, so the communicators have one rank; the two-rank layout of the collective
is covered on CPU by tests/test_dist_gloo.py.  Bar: idx, d1, d2 bit-exact against the oracle's serial
ascending scan (src/ORBmatcher.cc:461-485)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
INT_MAX = 2147483647


def _planted(synth, nq, nt, seed):
    q, t = synth.synth_descriptors(nq, nt, seed=seed)
    t[nt - 50:nt - 10] = t[10]  # duplicates far apart: the first index must win
    q[0] = t[10]
    q[1] = t[nt - 1]
    return q, t


def test_knn2_sharded_torch_form_world1(orb, orbref, synth):
    import torch
    nq, nt, base = 777, 30011, 123456
    q, t = _planted(synth, nq, nt, 31)
    m = orb.ORBmatcher(0.6, True)
    dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
    parts_local = torch.full((3, nq), -3, dtype=torch.int32, device="cuda")
    parts_all = torch.full((1, 3, nq), -4, dtype=torch.int32, device="cuda")
    out = torch.full((3, nq), -5, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    orb.sharded.knn2_sharded(m, dq, nq, dt, nt, base, parts_local, parts_all, out)
    ridx, rd1, rd2 = orbref.knn2(q, t)
    o = out.cpu().numpy()
    assert np.array_equal(o[0], ridx + base) and np.array_equal(o[1], rd1) and np.array_equal(o[2], rd2)
    assert o[0, 0] == base + 10 and o[1, 0] == 0 and o[2, 0] == 0


def test_knn2_sharded_c_entry_with_rccl_communicator(orb, orbref, synth):
    """orbhip_comm_create (ncclCommInitRank, one rank) + orbhip_knn2_sharded: scan, in-place ncclAllGather
    and merge on the matcher's stream.  Two calls: the second with fewer queries reuses the gather buffer."""
    import torch
    comm = orb.sharded.Communicator(orb.sharded.Communicator.unique_id(), 0, 1, 0)
    L = orb._lib.lib()
    assert L.orbhip_comm_world(comm._h) == 1 and L.orbhip_comm_rank(comm._h) == 0
    m = orb.ORBmatcher(0.6, True)
    for nq, nt, base, seed in ((1500, 70001, 5000000, 41), (300, 999, 0, 42)):
        q, t = _planted(synth, nq, nt, seed)
        dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
        out = torch.full((3, nq), -5, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        orb.sharded.knn2_sharded_rccl(m, comm, dq, nq, dt, nt, base, out)
        m.synchronize()
        ridx, rd1, rd2 = orbref.knn2(q, t)
        o = out.cpu().numpy()
        assert np.array_equal(o[0], ridx + base) and np.array_equal(o[1], rd1) and np.array_equal(o[2], rd2)
    # no communicator = single shard; an empty shard gives the reference's INT_MAX initialisation
    out = torch.full((3, 300), -5, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    orb.sharded.knn2_sharded_rccl(m, None, dq, 300, dt, 999, 7, out)
    m.synchronize()
    assert np.array_equal(out[0].cpu().numpy(), ridx + 7)
    orb.sharded.knn2_sharded_rccl(m, comm, dq, 300, dt, 0, 0, out)
    m.synchronize()
    o = out.cpu().numpy()
    assert (o[0] == -1).all() and (o[1] == INT_MAX).all() and (o[2] == INT_MAX).all()
    comm.close()


def test_full_size_map_shards_merged_on_device(orb, orbref):
    """Two shards of the frame-versus-map configuration at full size (4000 queries x 1.25 M rows per GPU),
    scanned one after the other into a [shards, 3, nq] device tensor with their global index bases and
    merged by the device merge; 256 seeded queries (planted exact and near duplicates within and across
    shards, far apart) are checked against the oracle's serial scan of all 2.5 M rows."""
    import torch
    nq, per, S = 4000, 1250000, 2
    nt = per * S
    rng = np.random.default_rng(77)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    sample = np.sort(rng.choice(nq, 256, replace=False))
    # planted rows: query s copies a map row (some with a few bits flipped); the row is duplicated later in
    # the same shard and in the other shard, so that "first index attaining the minimum" and the
    # multiplicity of the second distance are exercised across the shard boundary
    for k, s in enumerate(sample[:96]):
        src = int(rng.integers(0, per))
        q[s] = t[src]
        if k % 3 == 0:
            q[s, rng.integers(0, 32)] ^= np.uint8(1 << int(rng.integers(0, 8)))
        t[(src + 600000 + k) % per] = t[src]
        if k % 2 == 0:
            t[per + int(rng.integers(0, per))] = t[src]
    # a best match that only exists in the second shard, and the very last row of the map
    q[sample[100]] = t[nt - 1]
    q[sample[101]] = t[per]
    m = orb.ORBmatcher(0.6, True)
    dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
    parts = torch.full((S, 3, nq), -3, dtype=torch.int32, device="cuda")
    out = torch.full((3, nq), -5, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for g in range(S):
        m.knn2_device(dq.data_ptr(), nq, 0, dt.data_ptr() + g * per * 32, per, 0, 1, g * per, parts[g, 0].data_ptr(),
                      parts[g, 1].data_ptr(), parts[g, 2].data_ptr())
    m.merge_device(parts.data_ptr(), S, nq, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
    m.synchronize()
    o = out.cpu().numpy()
    ridx, rd1, rd2 = orbref.knn2(np.ascontiguousarray(q[sample]), t)
    assert np.array_equal(o[0, sample], ridx) and np.array_equal(o[1, sample], rd1) and np.array_equal(o[2, sample], rd2)
    assert o[0, sample[100]] <= nt - 1 and o[1, sample[100]] == 0
    assert (o[1, sample[:96]] <= 1).all() and (o[2, sample[:96]] <= 1).sum() >= 30  # duplicates were found
    # the unsharded scan of the same 2.5 M rows gives the same answer for every query
    whole = torch.full((3, nq), -6, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    m.knn2_device(dq.data_ptr(), nq, 0, dt.data_ptr(), nt, 0, 1, 0, whole[0].data_ptr(), whole[1].data_ptr(),
                  whole[2].data_ptr())
    m.synchronize()
    assert np.array_equal(whole.cpu().numpy(), o)
