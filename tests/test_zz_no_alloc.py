"""The entry point that allocates nothing (orbhip_knn2_device_noalloc) against the existing
orbhip_knn2_device, through the C ABI: two calls in a row on one matcher, the second with a smaller
input, so that stale partial results of the first call in the GPU's shared scratch array would show.
Bar: idx, d1, d2 bit-exact (integer work).  Also checked against the oracle, and that a call whose
partial results exceed the fixed array is rejected with ORBHIP_ECAP."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(m, torch, q, t, idx_base):
    nq, nt = len(q), len(t)
    dq = torch.from_numpy(q).cuda()
    dt = torch.from_numpy(t).cuda()
    a = torch.full((3, nq), -7, dtype=torch.int32, device="cuda")
    b = torch.full((3, nq), -9, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    m.knn2_device(dq.data_ptr(), nq, nq * 32, dt.data_ptr(), nt, nt * 32, 1, idx_base,
                  a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr())
    m.knn2_device_noalloc(dq.data_ptr(), nq, nq * 32, dt.data_ptr(), nt, nt * 32, 1, idx_base,
                          b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr())
    m.synchronize()
    return a.cpu().numpy(), b.cpu().numpy()


def test_noalloc_matches_existing_entry_point_twice(orb, orbref, synth):
    import torch
    m = orb.ORBmatcher(0.6, True)
    # first call: 3000 queries x 60000 rows -> many train chunks, partial results in the scratch array
    q, t = synth.synth_descriptors(3000, 60000, seed=21)
    t[50000:50040] = t[7]
    q[0] = t[7]
    a, b = _both(m, torch, q, t, 1000)
    assert np.array_equal(a, b)
    assert b[0, 0] == 1007 and b[1, 0] == 0 and b[2, 0] == 0
    ridx, rd1, rd2 = orbref.knn2(q[:64], t)
    assert np.array_equal(b[0, :64], ridx + 1000) and np.array_equal(b[1, :64], rd1) and np.array_equal(b[2, :64], rd2)
    # second call on the same matcher and GPU array: smaller input, other chunking
    q2, t2 = synth.synth_descriptors(300, 5000, seed=22)
    a2, b2 = _both(m, torch, q2, t2, 0)
    assert np.array_equal(a2, b2)
    ridx, rd1, rd2 = orbref.knn2(q2, t2)
    assert np.array_equal(b2[0], ridx) and np.array_equal(b2[1], rd1) and np.array_equal(b2[2], rd2)
    # a second matcher on the same GPU shares the array (sequential use is fine)
    m2 = orb.ORBmatcher(0.6, True)
    a3, b3 = _both(m2, torch, q2[:1], t2[:1], 5)
    assert np.array_equal(a3, b3) and b3[0, 0] == 5 and b3[2, 0] == 2147483647


def test_noalloc_rejects_oversized_partials(orb):
    """The plan keeps partial results under 8 MiB unless a chunk would exceed the 22-bit row field of the
    packed key: 1 M queries against 9 M rows need 3 chunks x 1 M x 8 B = 24 MB > the 16 MiB array.  The call
    must fail with ORBHIP_ECAP before launching anything; the growable entry point accepts the shape."""
    import torch
    L = orb._lib.lib()
    assert L.orbhip_knn2_scratch_bytes() == 16 << 20
    m = orb.ORBmatcher()
    nq, nt = 1000000, 9000000
    dq = torch.zeros((nq, 32), dtype=torch.uint8, device="cuda")
    dt = torch.zeros((nt, 32), dtype=torch.uint8, device="cuda")
    o = torch.full((3, nq), -5, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with pytest.raises(orb._lib.OrbhipError) as ei:
        m.knn2_device_noalloc(dq.data_ptr(), nq, nq * 32, dt.data_ptr(), nt, nt * 32, 1, 0, o[0].data_ptr(),
                              o[1].data_ptr(), o[2].data_ptr())
    assert ei.value.code == orb._lib.ECAP and "scratch" in str(ei.value)
    m.synchronize()
    assert int(o.max()) == -5 and int(o.min()) == -5  # nothing ran
    # a shape without partial results (one chunk) passes whatever its size
    nt = 1024
    m.knn2_device_noalloc(dq.data_ptr(), nq, nq * 32, dt.data_ptr(), nt, nt * 32, 1, 0, o[0].data_ptr(), o[1].data_ptr(),
                          o[2].data_ptr())
    m.synchronize()
    assert int(o[0].max()) == 0 and int(o[1].max()) == 0 and int(o[2].max()) == 0


def test_noalloc_calls_of_two_matchers_are_ordered_on_the_device(orb, synth):
    """Two handles (two streams) on one GPU call back to back with no host synchronisation in between, several
    rounds, every call with several train chunks (partial results in the shared array).  Each call waits on the
    device for the previous user of the array, so every result equals the growable entry point's."""
    import torch
    L = orb._lib.lib()
    ma, mb = orb.ORBmatcher(0.6, True), orb.ORBmatcher(0.6, True)
    shapes = [(1500, 200000, 31), (700, 90000, 32)]
    data, want = [], []
    for m, (nq, nt, seed) in zip((ma, mb), shapes):
        q, t = synth.synth_descriptors(nq, nt, seed=seed)
        dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
        assert L.orbhip_knn2_partial_bytes(m._h, nq, nt, 1) > 0  # several chunks: the array is used
        ref = torch.full((3, nq), -7, dtype=torch.int32, device="cuda")
        m.knn2_device(dq.data_ptr(), nq, nq * 32, dt.data_ptr(), nt, nt * 32, 1, 0, ref[0].data_ptr(), ref[1].data_ptr(), ref[2].data_ptr())
        m.synchronize()
        data.append((dq, dt, nq, nt))
        want.append(ref.cpu().numpy())
    rounds = 6
    outs = [[torch.full((3, d[2]), -9, dtype=torch.int32, device="cuda") for _ in range(rounds)] for d in data]
    torch.cuda.synchronize()
    for r in range(rounds):
        for k, m in enumerate((ma, mb)):
            dq, dt, nq, nt = data[k]
            o = outs[k][r]
            m.knn2_device_noalloc(dq.data_ptr(), nq, nq * 32, dt.data_ptr(), nt, nt * 32, 1, 0, o[0].data_ptr(), o[1].data_ptr(),
                                  o[2].data_ptr())
    ma.synchronize()
    mb.synchronize()
    for k in range(2):
        for r in range(rounds):
            assert np.array_equal(outs[k][r].cpu().numpy(), want[k]), "matcher %d, round %d" % (k, r)
    # the byte count the library reports is what decides ECAP: the rejected shape of the test above needs more than the array
    assert L.orbhip_knn2_partial_bytes(ma._h, 1000000, 9000000, 1) > L.orbhip_knn2_scratch_bytes()
    assert L.orbhip_knn2_partial_bytes(ma._h, 1000000, 1024, 1) == 0
    assert L.orbhip_knn2_partial_bytes(ma._h, -1, 5, 1) == -1
