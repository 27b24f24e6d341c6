"""Parity of the device feature grid, GetFeaturesInArea and the two windowed matchers
(SearchForInitialization, WindowSearch) against the oracle's literal restatements
(oracle/frame_ref.cpp), through the C ABI.  Bar: bit-exact (index / integer work; the float
expressions involved are evaluated operation by operation in fp32 on both sides)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
INT_MAX = 2147483647


def make_frame(orbref, rng, n, w, h, levels=3, pool=None, cluster=False):
    kp = np.zeros(n, orbref.KP_DTYPE)
    if cluster:  # crowded: windows overlap heavily, many keypoints compete for the same features
        cx, cy = rng.uniform(0.2 * w, 0.8 * w, 8), rng.uniform(0.2 * h, 0.8 * h, 8)
        k = rng.integers(0, 8, n)
        kp["x"] = np.clip(cx[k] + rng.normal(0, 25, n), -3, w + 3).astype(np.float32)
        kp["y"] = np.clip(cy[k] + rng.normal(0, 25, n), -3, h + 3).astype(np.float32)
    else:
        kp["x"] = rng.uniform(-2, w + 2, n).astype(np.float32)
        kp["y"] = rng.uniform(-2, h + 2, n).astype(np.float32)
    # a share of integer and half-integer coordinates: exact cell / window boundaries
    m = rng.random(n) < 0.3
    kp["x"][m] = np.round(kp["x"][m] * 2) / 2
    kp["y"][m] = np.round(kp["y"][m] * 2) / 2
    kp["octave"] = rng.integers(0, levels, n)
    kp["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    kp["size"] = 31.0
    kp["class_id"] = -1
    if pool is None:
        desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    else:  # few distinct descriptors + small perturbations: heavy distance ties
        desc = pool[rng.integers(0, len(pool), n)].copy()
        flips = rng.integers(0, 4, n)
        for i in range(n):
            for _ in range(flips[i]):
                desc[i, rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
    return kp, desc


def moved(orbref, rng, kp, desc, w, h, drop=0.1, noise_bits=6, shift=6.0, rot=None):
    """A second view of a frame: keypoints jittered, some dropped, descriptors perturbed, shuffled."""
    n = len(kp)
    keep = rng.random(n) > drop
    kp2, d2 = kp[keep].copy(), desc[keep].copy()
    kp2["x"] += rng.uniform(-shift, shift, len(kp2)).astype(np.float32)
    kp2["y"] += rng.uniform(-shift, shift, len(kp2)).astype(np.float32)
    if rot is not None:
        kp2["angle"] = np.mod(kp2["angle"] + np.float32(rot) + rng.normal(0, 8, len(kp2)).astype(np.float32), 360).astype(np.float32)
        wild = rng.random(len(kp2)) < 0.15
        kp2["angle"][wild] = rng.uniform(0, 360, wild.sum()).astype(np.float32)
    for i in range(len(kp2)):
        for _ in range(rng.integers(0, noise_bits + 1)):
            d2[i, rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
    p = rng.permutation(len(kp2))
    return kp2[p], d2[p]


@pytest.fixture(scope="module")
def matcher(orb):
    return orb.ORBmatcher(0.9, True)


def frames_for(orbref, seed, sizes, w, h, **kw):
    rng = np.random.default_rng(seed)
    pool = rng.integers(0, 256, (12, 32), dtype=np.uint8) if kw.pop("ties", False) else None
    return [make_frame(orbref, rng, n, w, h, pool=pool, **kw) for n in sizes], rng


@pytest.mark.parametrize("w,h", [(640, 480), (1280, 720), (752, 480)])
def test_frame_grid(orb, orbref, matcher, w, h):
    fr, _ = frames_for(orbref, 5, [0, 1, 63, 64, 65, 500, 2000, 4000], w, h)
    bounds = [0, 0, w, h]
    fb = orb.FrameBatch.from_host(matcher, [k for k, _ in fr], [d for _, d in fr], bounds)
    for f, (kp, _) in enumerate(fr):
        rs, ri = orbref.frame_grid(kp, bounds)
        ds, di = fb.grid(f)
        assert np.array_equal(rs, ds), "frame %d cell starts" % f
        assert np.array_equal(ri, di), "frame %d cell items" % f


def test_frame_grid_distorted_bounds(orb, orbref, matcher):
    """Bounds of an undistorted image need not start at 0 (src/Frame.cc:337-340)."""
    fr, _ = frames_for(orbref, 6, [800, 300], 640, 480)
    bounds = [-7, -4, 655, 489]
    fb = orb.FrameBatch.from_host(matcher, [k for k, _ in fr], [d for _, d in fr], bounds)
    for f, (kp, _) in enumerate(fr):
        rs, ri = orbref.frame_grid(kp, bounds)
        ds, di = fb.grid(f)
        assert np.array_equal(rs, ds) and np.array_equal(ri, di)


def test_features_in_area(orb, orbref, matcher):
    w, h = 640, 480
    fr, rng = frames_for(orbref, 7, [1500, 40, 0], w, h, levels=4)
    bounds = [0, 0, w, h]
    fb = orb.FrameBatch.from_host(matcher, [k for k, _ in fr], [d for _, d in fr], bounds)
    nq = 300
    q = np.zeros((3, nq), orb.AREA_QUERY_DTYPE)
    q["x"] = rng.uniform(-150, w + 150, (3, nq)).astype(np.float32)
    q["y"] = rng.uniform(-150, h + 150, (3, nq)).astype(np.float32)
    q["r"] = rng.choice(np.array([0.0, 1.0, 7.5, 10.0, 50.0, 100.0, 200.0, 2000.0], np.float32), (3, nq))
    kind = rng.integers(0, 3, (3, nq))
    lo = rng.integers(0, 4, (3, nq))
    q["min_level"] = np.where(kind == 0, -1, lo)
    q["max_level"] = np.where(kind == 0, -1, np.where(kind == 1, lo, lo + rng.integers(0, 3, (3, nq))))
    # exact hits: centred on keypoints, integer radii
    q["x"][0, :50] = fr[0][0]["x"][:50]
    q["y"][0, :50] = fr[0][0]["y"][:50]
    starts, out = fb.GetFeaturesInArea(q, out_cap=nq * 1500)
    total = 0
    for f, (kp, _) in enumerate(fr):
        for i in range(nq):
            ref = orbref.features_in_area(kp, bounds, float(q["x"][f, i]), float(q["y"][f, i]), float(q["r"][f, i]),
                                          int(q["min_level"][f, i]), int(q["max_level"][f, i]))
            dev = out[f, starts[f, i]:starts[f, i + 1]]
            assert np.array_equal(ref, dev), "frame %d query %d" % (f, i)
            total += len(ref)
    assert total > 10000


def test_features_in_area_overflow_is_reported(orb, orbref, matcher):
    fr, rng = frames_for(orbref, 8, [500], 640, 480)
    fb = orb.FrameBatch.from_host(matcher, [fr[0][0]], [fr[0][1]], [0, 0, 640, 480])
    q = np.zeros((1, 4), orb.AREA_QUERY_DTYPE)
    q["x"], q["y"], q["r"], q["min_level"], q["max_level"] = 320, 240, 1000, -1, -1
    with pytest.raises(orb.OrbhipError):
        fb.GetFeaturesInArea(q, out_cap=100)


def _run_search_init(orb, orbref, matcher, pairs, bounds, window, ratio, ori, cap=None):
    import torch
    m = orb.ORBmatcher(ratio, ori)
    F1 = orb.FrameBatch.from_host(m, [p[0] for p in pairs], [p[1] for p in pairs], bounds, cap=cap)
    F2 = orb.FrameBatch.from_host(m, [p[2] for p in pairs], [p[3] for p in pairs], bounds, cap=F1.cap if cap is None else cap)
    assert F1.cap == F2.cap
    prev = np.zeros((len(pairs), F1.cap, 2), np.float32)
    for f, p in enumerate(pairs):
        prev[f, :len(p[0]), 0], prev[f, :len(p[0]), 1] = p[0]["x"], p[0]["y"]
    d_prev = torch.from_numpy(prev.copy()).cuda()
    nm, m12 = m.SearchForInitialization(F1, F2, d_prev, window)
    nm, m12, prev_out = nm.cpu().numpy(), m12.cpu().numpy(), d_prev.cpu().numpy()
    matched = 0
    for f, p in enumerate(pairs):
        n1 = len(p[0])
        rn, r12, rprev = orbref.search_for_initialization(p[0], p[1], p[2], p[3], bounds, prev[f, :n1], window,
                                                          m.mfNNratio, ori)
        assert np.array_equal(r12, m12[f, :n1]), "pair %d vnMatches12 (%d differ)" % (f, int((r12 != m12[f, :n1]).sum()))
        assert (m12[f, n1:] == -1).all()
        assert rn == nm[f], "pair %d nmatches" % f
        assert np.array_equal(rprev.view(np.uint32), prev_out[f, :n1].view(np.uint32)), "pair %d vbPrevMatched" % f
        matched += rn
    return matched


def _pad_equal(pairs):
    cap = max(max(len(p[0]), len(p[2])) for p in pairs)
    return max(cap, 1)


@pytest.mark.parametrize("ties,cluster", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("ori", [True, False])
def test_search_for_initialization(orb, orbref, matcher, ties, cluster, ori):
    w, h = 640, 480
    fr, rng = frames_for(orbref, 11 + ties + 2 * cluster, [0, 1, 70, 900, 2000, 2500], w, h, levels=2, ties=ties,
                         cluster=cluster)
    pairs = []
    for kp, de in fr:
        kp2, de2 = moved(orbref, rng, kp, de, w, h, rot=17.0)
        pairs.append((kp, de, kp2, de2))
    pairs.append((fr[3][0], fr[3][1], fr[3][0][:0], fr[3][1][:0]))  # empty F2
    matched = _run_search_init(orb, orbref, matcher, pairs, [0, 0, w, h], 100, 0.9, ori, cap=_pad_equal(pairs))
    assert matched > 300


def test_search_for_initialization_small_window_strict_ratio(orb, orbref, matcher):
    w, h = 1280, 720
    fr, rng = frames_for(orbref, 21, [3000, 4000], w, h, levels=1, ties=True, cluster=True)
    pairs = [(kp, de) + moved(orbref, rng, kp, de, w, h, noise_bits=2, shift=3.0, rot=200.0) for kp, de in fr]
    assert _run_search_init(orb, orbref, matcher, pairs, [0, 0, w, h], 10, 0.6, True, cap=_pad_equal(pairs)) > 50
    assert _run_search_init(orb, orbref, matcher, pairs, [0, 0, w, h], 200, 0.95, True, cap=_pad_equal(pairs)) > 50


def _run_window(orb, orbref, pairs, valids, bounds, window, lo, hi, ratio, ori):
    import torch
    m = orb.ORBmatcher(ratio, ori)
    cap = _pad_equal(pairs)
    F1 = orb.FrameBatch.from_host(m, [p[0] for p in pairs], [p[1] for p in pairs], bounds, cap=cap)
    F2 = orb.FrameBatch.from_host(m, [p[2] for p in pairs], [p[3] for p in pairs], bounds, cap=cap)
    v = np.zeros((len(pairs), cap), np.uint8)
    for f, val in enumerate(valids):
        v[f, :len(val)] = val
    nm, m21 = m.WindowSearch(F1, F2, window, torch.from_numpy(v).cuda(), lo, hi)
    nm, m21 = nm.cpu().numpy(), m21.cpu().numpy()
    matched = 0
    for f, p in enumerate(pairs):
        n2 = len(p[2])
        rn, r21 = orbref.window_search(p[0], p[1], valids[f], p[2], p[3], bounds, window, lo, hi, m.mfNNratio, ori)
        assert np.array_equal(r21, m21[f, :n2]), "pair %d vnMatches21 (%d differ)" % (f, int((r21 != m21[f, :n2]).sum()))
        assert (m21[f, n2:] == -1).all()
        assert rn == nm[f], "pair %d nmatches" % f
        matched += rn
    return matched


@pytest.mark.parametrize("window,lo,hi", [(200, -1, INT_MAX), (100, 0, INT_MAX), (50, 2, INT_MAX), (100, 1, 2)])
@pytest.mark.parametrize("ori", [True, False])
def test_window_search(orb, orbref, window, lo, hi, ori):
    w, h = 640, 480
    fr, rng = frames_for(orbref, 31, [0, 5, 64, 1000, 2000], w, h, levels=4, ties=True, cluster=True)
    pairs, valids = [], []
    for kp, de in fr:
        pairs.append((kp, de) + moved(orbref, rng, kp, de, w, h, rot=300.0))
        valids.append((rng.random(len(kp)) < 0.7).astype(np.uint8))
    matched = _run_window(orb, orbref, pairs, valids, [0, 0, w, h], window, lo, hi, 0.6, ori)
    assert matched > 100


def test_window_search_on_extracted_frames(orb, orbref, synth):
    """End to end on the extractor's own output: two synthetic frames, the second a shifted copy."""
    w, h = 640, 480
    ext = orb.ORBextractor(1000, 1.2, 8, 1, 20, max_width=w, max_height=h, max_batch=2)
    a = synth.synth_image(w, h, 3)
    b = np.roll(a, (4, -6), axis=(0, 1))
    (ka, da), (kb, db) = ext(a), ext(b)
    pairs = [(ka, da, kb, db), (kb, db, ka, da)]
    valids = [np.ones(len(ka), np.uint8), np.ones(len(kb), np.uint8)]
    assert _run_window(orb, orbref, pairs, valids, [0, 0, w, h], 30, -1, INT_MAX, 0.8, True) > 200
    m = orb.ORBmatcher(0.9, True)
    assert _run_search_init(orb, orbref, m, pairs, [0, 0, w, h], 100, 0.9, True, cap=_pad_equal(pairs)) > 50


def test_host_pointer_forms(orb, orbref):
    """The single-pair host forms (copy in, grid, match, copy out) give the oracle's answers."""
    w, h = 752, 480
    fr, rng = frames_for(orbref, 41, [1200, 0, 3], w, h, levels=3, ties=True, cluster=True)
    bounds = [0, 0, w, h]
    for kp, de in fr:
        kp2, de2 = moved(orbref, rng, kp, de, w, h, rot=45.0)
        m = orb.ORBmatcher(0.9, True)
        prev = np.stack([kp["x"], kp["y"]], 1).astype(np.float32).reshape(-1, 2)
        n, m12, pv = m.search_for_initialization_host(kp, de, kp2, de2, bounds, prev, 100)
        rn, r12, rpv = orbref.search_for_initialization(kp, de, kp2, de2, bounds, prev, 100, m.mfNNratio, True)
        assert n == rn and np.array_equal(m12, r12) and np.array_equal(pv.view(np.uint32), rpv.view(np.uint32))
        m = orb.ORBmatcher(0.6, True)
        valid = (rng.random(len(kp)) < 0.8).astype(np.uint8)
        n, m21 = m.window_search_host(kp, de, valid, kp2, de2, bounds, 100, 0)
        rn, r21 = orbref.window_search(kp, de, valid, kp2, de2, bounds, 100, 0, INT_MAX, m.mfNNratio, True)
        assert n == rn and np.array_equal(m21, r21)


def test_distinctive_descriptors(orb, orbref):
    """MapPoint::ComputeDistinctiveDescriptors, many map points at once: BestIdx and BestMedian."""
    import torch
    rng = np.random.default_rng(51)
    sizes = [0, 1, 2, 3, 4, 5, 7, 8, 33, 64, 65, 100, 128] + list(rng.integers(1, 40, 300))
    descs, start = [], [0]
    for n in sizes:
        base = rng.integers(0, 256, (1, 32), dtype=np.uint8)
        d = np.repeat(base, n, axis=0)
        for i in range(n):  # observations of one point: a common descriptor with a few flipped bits (ties abound)
            for _ in range(rng.integers(0, 12)):
                d[i, rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
        if n >= 3 and rng.random() < 0.3:
            d[rng.integers(0, n)] = ~base[0]  # an outlier at distance ~256
        descs.append(d)
        start.append(start[-1] + n)
    alld = np.concatenate(descs, 0)
    m = orb.ORBmatcher()
    idx, med = m.ComputeDistinctiveDescriptors(torch.from_numpy(alld).cuda(), torch.tensor(start, dtype=torch.int32).cuda())
    idx, med = idx.cpu().numpy(), med.cpu().numpy()
    for p, d in enumerate(descs):
        ri, rm = orbref.distinctive_descriptor(d)
        assert (idx[p], med[p]) == (ri, rm), "point %d (N = %d)" % (p, len(d))
    # more observations than the LDS matrix holds are reported, not silently truncated
    big = rng.integers(0, 256, (129, 32), dtype=np.uint8)
    with pytest.raises(orb.OrbhipError):
        m.ComputeDistinctiveDescriptors(torch.from_numpy(big).cuda(), torch.tensor([0, 129], dtype=torch.int32).cuda())


def test_candidate_lists_longer_than_the_lds_run_buffer(orb, orbref, matcher):
    """5000 keypoints of one level inside a 200x150 frame and a window that covers the frame: every
    candidate list holds all 5000 features, more than the 4096 entries the resolution kernels stage in
    LDS, so they read the lists from global memory."""
    w, h = 200, 150
    fr, rng = frames_for(orbref, 61, [5000, 4500], w, h, levels=1, ties=True)
    pairs, valids = [], []
    for kp, de in fr:
        pairs.append((kp, de) + moved(orbref, rng, kp, de, w, h, drop=0.0, rot=10.0))
        valids.append((rng.random(len(kp)) < 0.9).astype(np.uint8))
    assert _run_search_init(orb, orbref, matcher, pairs, [0, 0, w, h], 300, 0.9, True, cap=_pad_equal(pairs)) > 100
    assert _run_window(orb, orbref, pairs, valids, [0, 0, w, h], 300, -1, INT_MAX, 0.8, True) > 100
