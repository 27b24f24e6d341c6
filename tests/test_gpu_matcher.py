"""Parity of the HIP Hamming kNN (k=2) against the oracle's scalar scan, through the C ABI.
Bar: idx, d1, d2 bit-exact (integer work)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
INT_MAX = 2147483647

# both scan kernels of orbhip_knn2*: the 4-bit matrix-core one (default) and the xor + popcount one
SCANS = ["matrix", "popcount"]


def _matcher(orb, scan, *args):
    m = orb.ORBmatcher(*args)
    m.set_scan_kernel(scan)
    return m


@pytest.mark.parametrize("scan", SCANS)
@pytest.mark.parametrize("nq,nt", [(1, 1), (1, 2), (7, 1), (5, 3), (256, 255), (257, 513), (2000, 2000), (300, 5000)])
def test_knn2_single_pair(orb, orbref, synth, nq, nt, scan):
    q, t = synth.synth_descriptors(nq, nt, seed=nq * 7919 + nt)
    m = _matcher(orb, scan, 0.6, True)
    idx, d1, d2 = m.knn2(q, t)
    ridx, rd1, rd2 = orbref.knn2(q, t)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(d1, rd1)
    assert np.array_equal(d2, rd2)


@pytest.mark.parametrize("scan", SCANS)
@pytest.mark.parametrize("nt", [8, 9, 15, 16, 31, 32, 33, 63, 64, 65, 511, 512, 513, 519, 520, 527, 1031, 4099])
def test_knn2_group_boundaries(orb, orbref, synth, nt, scan):
    """Train sizes around the scans' internal boundaries: groups of 8 rows, the 512 rows folded unconditionally
    before the threshold filter starts and the rows left over after the last whole group (popcount kernel); tiles
    of 32 rows and the partial last tile (matrix kernel)."""
    q, t = synth.synth_descriptors(300, nt, seed=4242 + nt)
    # the best rows of some queries sit at the very end and right after the unconditional part
    q[0] = t[nt - 1]
    q[1] = t[nt - 1] ^ np.uint8(1)
    if nt > 513:
        q[2] = t[512]
        t[513] = t[512]
        q[3] = t[nt - 2]
        t[nt - 1] = t[nt - 2]
    m = _matcher(orb, scan, 0.6, True)
    idx, d1, d2 = m.knn2(q, t)
    ridx, rd1, rd2 = orbref.knn2(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(d1, rd1) and np.array_equal(d2, rd2)


@pytest.mark.parametrize("scan", SCANS)
def test_knn2_late_improvements(orb, orbref, scan):
    """Distances that keep falling along the scan (row j differs from the query in about 200 - j/20 bits), so the
    filtered steady state takes its exact-fold path again and again, with equal distances in neighbouring rows."""
    rng = np.random.default_rng(77)
    nq, nt = 130, 3000
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    base = q[rng.integers(0, nq, nt)]
    bits = np.zeros((nt, 256), np.uint8)
    for j in range(nt):
        k = max(0, 200 - j // 20)
        bits[j, rng.permutation(256)[:k]] = 1
    t = base ^ np.packbits(bits, axis=1)
    m = _matcher(orb, scan, 0.6, True)
    idx, d1, d2 = m.knn2(q, t)
    ridx, rd1, rd2 = orbref.knn2(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(d1, rd1) and np.array_equal(d2, rd2)


@pytest.mark.parametrize("scan", SCANS)
def test_knn2_empty_train(orb, scan):
    m = _matcher(orb, scan)
    q = np.zeros((5, 32), np.uint8)
    idx, d1, d2 = m.knn2(q, np.zeros((0, 32), np.uint8))
    assert (idx == -1).all() and (d1 == INT_MAX).all() and (d2 == INT_MAX).all()


@pytest.mark.parametrize("scan", SCANS)
def test_knn2_ties_take_first_index(orb, orbref, scan):
    """All train rows identical: best = row 0, second = the same distance (multiplicity)."""
    q = np.random.default_rng(0).integers(0, 256, (70, 32), dtype=np.uint8)
    t = np.tile(np.arange(32, dtype=np.uint8), (900, 1))
    m = _matcher(orb, scan)
    idx, d1, d2 = m.knn2(q, t)
    assert (idx == 0).all() and np.array_equal(d1, d2)
    ridx, rd1, rd2 = orbref.knn2(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(d1, rd1) and np.array_equal(d2, rd2)


@pytest.mark.parametrize("scan", SCANS)
def test_knn2_extreme_popcounts(orb, orbref, scan):
    """Descriptors of all zeros / all ones and their neighbours: distances 0, 1, 255 and 256, query popcounts 0 and
    256 (the matrix kernel carries distance - popcount(query) through its accumulators), equal distances on both
    sides of a tile boundary."""
    rng = np.random.default_rng(5)
    nt = 200
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    t[3] = 0
    t[40] = 0xff
    t[41] = 0xff
    t[41, 7] = 0x7f
    t[31] = t[32] = t[17]
    t[63] = t[64] = t[100] ^ np.uint8(0x10)
    q = rng.integers(0, 256, (100, 32), dtype=np.uint8)
    q[0] = 0
    q[1] = 0xff
    q[2] = 0
    q[2, 0] = 1
    q[3] = t[17]
    q[4] = t[100]
    q[5] = ~t[3]
    m = _matcher(orb, scan)
    idx, d1, d2 = m.knn2(q, t)
    ridx, rd1, rd2 = orbref.knn2(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(d1, rd1) and np.array_equal(d2, rd2)
    assert d1[0] == 0 and idx[0] == 3 and d1[1] == 0 and idx[1] == 40 and d2[1] == 1 and idx[3] == 17 and d2[3] == 0
    # only all-zero / all-one rows: every distance is 0 or 256
    t2 = np.zeros((70, 32), np.uint8)
    t2[1::2] = 0xff
    idx, d1, d2 = m.knn2(q[:2], t2)
    ridx, rd1, rd2 = orbref.knn2(q[:2], t2)
    assert np.array_equal(idx, ridx) and np.array_equal(d1, rd1) and np.array_equal(d2, rd2)
    assert list(idx) == [0, 1] and list(d1) == [0, 0] and list(d2) == [0, 0]
    idx, d1, d2 = m.knn2(q[:2], t2[:2])
    assert list(d2) == [256, 256]


@pytest.mark.parametrize("scan", SCANS)
@pytest.mark.parametrize("nt", [736, 767, 768, 769, 799, 800, 1023, 1024, 1025, 2000])
def test_knn2_insertion_phase_boundary(orb, orbref, nt, scan):
    """The matrix kernel keeps the two smallest of a chunk's first 768 rows (24 tiles) by sorted insertion on float
    keys (distance - popcount(query) + row / 1024) and the rest by votes and integer keys: equal rows on both sides
    of the hand-over, the best row as the last one of the first phase and the first one of the second, all-zero /
    all-one descriptors (keys at both ends of the float range) at the highest rows of the first phase, and queries
    whose two smallest both lie in one phase or one in each."""
    rng = np.random.default_rng(9000 + nt)
    q = rng.integers(0, 256, (320, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    last1 = min(767, nt - 1)           # last row of the insertion phase
    t[last1] = q[0]
    t[5] = q[1]
    if nt > 768:
        t[768] = q[1]                  # first row of the vote phase: the same distance as row 5, the lower row wins
    t[last1 - 1] = q[2] ^ np.uint8(1)  # distance 8 in the first phase ...
    if nt > 790:
        t[790] = q[2]                  # ... beaten in the second
        t[791] = q[2]                  # best and second best both after the hand-over, equal
    t[last1 - 2] = 0
    t[last1 - 3] = 0xff
    q[3] = 0
    q[4] = 0xff
    q[5] = 0xff
    q[5, 31] = 0xfe
    q[6] = t[0]                        # row 0 and row 1 equal: second best = same distance
    t[1] = t[0]
    q[7] = t[nt - 1]                   # the very last row
    m = _matcher(orb, scan)
    idx, d1, d2 = m.knn2(q, t)
    ridx, rd1, rd2 = orbref.knn2(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(d1, rd1) and np.array_equal(d2, rd2)
    assert d1[0] == 0 and d1[3] == 0 and d1[4] == 0 and d1[6] == 0 and d2[6] == 0 and idx[6] == 0
    assert idx[1] == 5 and d1[1] == 0 and (nt <= 768 or d2[1] == 0)


@pytest.mark.parametrize("scan", SCANS)
def test_knn2_batched_pairs(orb, orbref, synth, scan):
    P, nq, nt = 9, 300, 400
    qs, ts = zip(*[synth.synth_descriptors(nq, nt, seed=100 + p) for p in range(P)])
    q, t = np.stack(qs), np.stack(ts)
    m = _matcher(orb, scan)
    idx, d1, d2 = m.knn2(q, t)
    for p in range(P):
        ridx, rd1, rd2 = orbref.knn2(q[p], t[p])
        assert np.array_equal(idx[p], ridx) and np.array_equal(d1[p], rd1) and np.array_equal(d2[p], rd2)


_PAIRS64 = {}


@pytest.mark.parametrize("scan", SCANS)
def test_knn2_64_pairs_of_2000_x_2000_in_one_call(orb, orbref, synth, scan):
    """The frame-pair configuration at its own size: 64 independent pairs of 2000 x 2000 descriptors (a quarter of
    the queries planted as near-copies of a train row, a twentieth of the train rows exact duplicates) through ONE
    call, every pair against the oracle's scan."""
    P, nq, nt = 64, 2000, 2000
    if not _PAIRS64:
        qs, ts = zip(*[synth.synth_descriptors(nq, nt, seed=9000 + p) for p in range(P)])
        _PAIRS64["q"], _PAIRS64["t"] = np.stack(qs), np.stack(ts)
        _PAIRS64["ref"] = [orbref.knn2(_PAIRS64["q"][p], _PAIRS64["t"][p]) for p in range(P)]
    q, t = _PAIRS64["q"], _PAIRS64["t"]
    m = _matcher(orb, scan)
    idx, d1, d2 = m.knn2(q, t)
    for p in range(P):
        ridx, rd1, rd2 = _PAIRS64["ref"][p]
        assert np.array_equal(idx[p], ridx) and np.array_equal(d1[p], rd1) and np.array_equal(d2[p], rd2), "pair %d" % p
    # the planted copies are found: at least a fifth of the queries lie within 40 bits of their best row
    assert (d1 <= 40).mean() > 0.2


@pytest.mark.parametrize("scan", SCANS)
def test_knn2_chunked_large_train(orb, orbref, synth, scan):
    """Few queries against many rows: the train set is split over workgroups and folded."""
    q, t = synth.synth_descriptors(64, 200000, seed=3)
    t[150000:150050] = t[10]  # duplicates far apart: the first index must win
    q[0] = t[10]
    m = _matcher(orb, scan)
    idx, d1, d2 = m.knn2(q, t)
    ridx, rd1, rd2 = orbref.knn2(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(d1, rd1) and np.array_equal(d2, rd2)
    assert idx[0] == 10 and d1[0] == 0 and d2[0] == 0


@pytest.mark.parametrize("scan", SCANS)
def test_knn2_one_chunk_longer_than_the_float_keys_row_field(orb, orbref, synth, scan):
    """So many (query block, pair) units that every workgroup scans the whole train set as ONE chunk, and a train set
    of 70,000 rows: the matrix-core scan's float keys hold 2^15 rows, so it works through the chunk in three epochs
    (the last with a partial tile).  Equal rows are planted on both sides of the epoch boundaries, exact copies of
    queries in every epoch: the first index must win across epochs, and a second-best of an earlier epoch must
    keep later rows of the same distance out."""
    import torch
    P, nq, nt = 1100, 64, 70000
    rng = np.random.default_rng(77)
    _, t = synth.synth_descriptors(8, nt, seed=31)
    q = rng.integers(0, 256, (P, nq, 32), dtype=np.uint8)
    # pair 0: query 0 equals rows in all three epochs, query 1 only a row of the last epoch, query 2 rows either
    # side of the first boundary, query 3 a near copy early and an exact copy late
    t[[100, 40000, 69990]] = q[0, 0]
    t[69999] = q[0, 1]
    t[[32767, 32768]] = q[0, 2]
    near = q[0, 3].copy()
    near[0] ^= 1
    t[5] = near
    t[66000] = q[0, 3]
    # pair 1: the two best rows of query 0 are equal-distance rows in different epochs
    a, b = q[1, 0].copy(), q[1, 0].copy()
    a[3] ^= 0x10
    b[7] ^= 0x01
    t[30000], t[50000] = a, b
    m = _matcher(orb, scan)
    dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
    out = torch.zeros((3, P, nq), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    m.knn2_device(dq.data_ptr(), nq, nq * 32, dt.data_ptr(), nt, 0, P, 0, out[0].data_ptr(), out[1].data_ptr(),
                  out[2].data_ptr())
    m.synchronize()
    o = out.cpu().numpy()
    for p_ in (0, 1, 2, 547, 1099):
        ridx, rd1, rd2 = orbref.knn2(q[p_], t)
        assert np.array_equal(o[0, p_], ridx) and np.array_equal(o[1, p_], rd1) and np.array_equal(o[2, p_], rd2), p_
    assert o[0, 0, 0] == 100 and o[1, 0, 0] == 0 and o[2, 0, 0] == 0
    assert o[0, 0, 1] == 69999 and o[1, 0, 1] == 0
    assert o[0, 0, 2] == 32767 and o[2, 0, 2] == 0
    assert o[0, 0, 3] == 66000 and o[1, 0, 3] == 0 and o[2, 0, 3] == 1
    assert o[0, 1, 0] == 30000 and o[1, 1, 0] == 1 and o[2, 1, 0] == 1


@pytest.mark.parametrize("scan", SCANS)
def test_knn2_train_rows_at_4_byte_alignment(orb, orbref, synth, scan):
    """The train set only has to be 4-byte aligned (the matrix kernel reads it in 16-byte pieces all the same)."""
    import torch
    nq, nt = 300, 1500
    q, t = synth.synth_descriptors(nq, nt, seed=23)
    ridx, rd1, rd2 = orbref.knn2(q, t)
    m = _matcher(orb, scan)
    dq = torch.from_numpy(q).cuda()
    for off in (4, 8, 12, 20):
        buf = torch.zeros(nt * 32 + 64, dtype=torch.uint8, device="cuda")
        buf[off:off + nt * 32] = torch.from_numpy(t.reshape(-1)).cuda()
        out = torch.zeros((3, nq), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        m.knn2_device(dq.data_ptr(), nq, nq * 32, buf.data_ptr() + off, nt, nt * 32, 1, 0, out[0].data_ptr(),
                      out[1].data_ptr(), out[2].data_ptr())
        m.synchronize()
        o = out.cpu().numpy()
        assert np.array_equal(o[0], ridx) and np.array_equal(o[1], rd1) and np.array_equal(o[2], rd2), off


def test_sharded_merge_and_ratio(orb, orbref, synth):
    """Index-contiguous shards merged on the device equal the unsharded scan; ratio test forms."""
    import torch
    nq, nt, G = 500, 4096, 4
    q, t = synth.synth_descriptors(nq, nt, seed=11)
    ridx, rd1, rd2 = orbref.knn2(q, t)
    m = orb.ORBmatcher(0.6)
    dq = torch.from_numpy(q).cuda()
    dt = torch.from_numpy(t).cuda()
    parts = torch.empty((G, 3, nq), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    sz = nt // G
    for g in range(G):
        m.knn2_device(dq.data_ptr(), nq, nq * 32, dt.data_ptr() + g * sz * 32, sz, sz * 32, 1, g * sz,
                      parts[g, 0].data_ptr(), parts[g, 1].data_ptr(), parts[g, 2].data_ptr())
    out = torch.empty((3, nq), dtype=torch.int32, device="cuda")
    acc = torch.empty((2, nq), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    m.merge_device(parts.data_ptr(), G, nq, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
    m.ratio_test_device(out[1].data_ptr(), out[2].data_ptr(), nq, 0, 100, acc[0].data_ptr())
    m.ratio_test_device(out[1].data_ptr(), out[2].data_ptr(), nq, 1, 20, acc[1].data_ptr())
    m.synchronize()
    o = out.cpu().numpy()
    assert np.array_equal(o[0], ridx) and np.array_equal(o[1], rd1) and np.array_equal(o[2], rd2)
    L = orbref.lib()
    a0 = np.array([L.orbref_ratio_accept(int(a), int(b), 0.6, 0) and a <= 100 for a, b in zip(rd1, rd2)], np.uint8)
    a1 = np.array([L.orbref_ratio_accept(int(a), int(b), 0.6, 1) and a <= 20 for a, b in zip(rd1, rd2)], np.uint8)
    acc = acc.cpu().numpy()
    assert np.array_equal(acc[0], a0) and np.array_equal(acc[1], a1)
    assert a0.sum() > 0


def test_descriptor_distance_host(orb, orbref):
    rng = np.random.default_rng(1)
    for _ in range(200):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert orb.ORBmatcher.DescriptorDistance(a, b) == orbref.hamming(a, b)
