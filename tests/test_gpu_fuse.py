"""Parity of the device search behind ORBmatcher::Fuse (src/ORBmatcher.cc:1027-1145, :1147-1276) and
ORBmatcher::SearchBySim3 (:1278-1516) against the oracle's literal restatements (oracle/frame_ref.cpp), through
the C ABI.  The oracle projects the map points as the reference does and returns the probe it built per map
point; the device receives exactly those probes, so the comparison covers KeyFrame::GetFeaturesInArea, the level
test, the Hamming distances and the first-of-the-least rule, per map point.  The gates the reference applies to
the result (<= TH_LOW, <= TH_HIGH, mutual agreement) are applied here to the device output and compared with the
oracle's return values.  Bar: bit-exact."""
import numpy as np
import pytest

from test_gpu_projection import SCALE, Scene, pose

pytestmark = pytest.mark.gpu

INT_MAX = 2147483647
TH_LOW, TH_HIGH = 20, 100


def device_best(orb, frames, bounds, probe_sets, desc_sets):
    """frames: [(kp, desc)], probe_sets / desc_sets: per frame the probes and their descriptors."""
    import torch
    m = orb.ORBmatcher(0.6, True)
    F = orb.FrameBatch.from_host(m, [f[0] for f in frames], [f[1] for f in frames], bounds)
    B, pcap = len(frames), max(max(len(p) for p in probe_sets), 1)
    probes = np.zeros((B, pcap), orb.PROBE_DTYPE)
    pdesc = np.zeros((B, pcap, 32), np.uint8)
    npr = np.zeros(B, np.int32)
    for f in range(B):
        k = len(probe_sets[f])
        npr[f] = k
        probes[f, :k] = probe_sets[f]
        pdesc[f, :k] = desc_sets[f]
    probes[:, :]["reserved"] = 0
    d_probes = torch.from_numpy(probes.view(np.uint8).reshape(B, pcap, 32).copy()).cuda()
    bi, bd = m.BestInRadius(F, d_probes, torch.from_numpy(pdesc).cuda(), torch.from_numpy(npr).cuda())
    m.synchronize()
    bi, bd = bi.cpu().numpy(), bd.cpu().numpy()
    for f in range(B):  # slots past the frame's probes
        assert (bi[f, npr[f]:] == -1).all() and (bd[f, npr[f]:] == INT_MAX).all()
    return [bi[f, :npr[f]] for f in range(B)], [bd[f, :npr[f]] for f in range(B)], m


def map_point_geometry(rng, s, T, scale=1.0):
    """Normals and distance ranges of the scene's map points as seen from the camera of pose T (scale: of a Sim3)."""
    n = len(s.level)
    R = T[:3, :3].astype(np.float64) / scale
    Ow = -R.T @ (T[:3, 3].astype(np.float64) / scale)
    PO = s.pos.astype(np.float64) - Ow
    dist = np.linalg.norm(PO, axis=1)
    normal = (PO / np.maximum(dist[:, None], 1e-9) + rng.normal(0, 0.5, (n, 3))).astype(np.float32)
    min_d = (dist / SCALE[s.level] * rng.uniform(0.8, 1.25, n)).astype(np.float32)
    max_d = (min_d * rng.choice(np.array([1.1, 1.2 ** 7.5], np.float32), n, p=[0.1, 0.9])).astype(np.float32)
    return normal, min_d, max_d


SIZES = [1500, 0, 1, 70, 3000]


@pytest.mark.parametrize("sim3,th", [(False, 3.0), (False, 2.5), (True, 4.0)])
@pytest.mark.parametrize("ties", [False, True])
def test_fuse(orb, orbref, sim3, th, ties):
    """th: the default 2.5 in LocalMapping::SearchInNeighbors (src/LocalMapping.cc:414,439, include/ORBmatcher.h:87),
    4 in LoopClosing::SearchAndFuse (src/LoopClosing.cc:598)."""
    rng = np.random.default_rng(171 + 2 * sim3 + ties)
    scenes, refs = [], []
    for n in SIZES:
        scale = 1.3 if sim3 else 1.0
        T = pose(rng, scale=scale)
        s = Scene(orbref, rng, n, 752, 480, T, ties=ties)
        normal, min_d, max_d = map_point_geometry(rng, s, T, scale)
        skip = (rng.random(n) < 0.1).astype(np.uint8)
        refs.append(orbref.fuse(s.kp, s.desc, s.bounds, s.K, SCALE, T, skip, s.pos, normal, min_d, max_d, s.mp_desc, th, sim3=sim3))
        scenes.append(s)
    bi, bd, _ = device_best(orb, [(s.kp, s.desc) for s in scenes], scenes[0].bounds, [r[3] for r in refs],
                            [s.mp_desc for s in scenes])
    fused = 0
    for f, r in enumerate(refs):
        assert np.array_equal(bi[f], r[1]), "frame %d: %d best indices differ" % (f, int((bi[f] != r[1]).sum()))
        assert np.array_equal(bd[f], r[2]), "frame %d: best distances" % f
        assert int((bd[f] <= TH_LOW).sum()) == r[0], "frame %d: nFused" % f
        fused += r[0]
    assert fused > 800


def test_fuse_host_form_and_shared_best(orb, orbref):
    """The one-keyframe host form, and probes that share their best feature (nothing is taken away)."""
    rng = np.random.default_rng(181)
    T = pose(rng)
    s = Scene(orbref, rng, 900, 640, 480, T, ties=True)
    normal, min_d, max_d = map_point_geometry(rng, s, T)
    # every map point twice: the copy must find the same feature
    dup = lambda a: np.concatenate([a, a])  # noqa: E731
    skip = np.zeros(2 * len(s.level), np.uint8)
    n, bi, bd, probes = orbref.fuse(s.kp, s.desc, s.bounds, s.K, SCALE, T, skip, dup(s.pos), dup(normal), dup(min_d), dup(max_d),
                                    dup(s.mp_desc), 3.0)
    m = orb.ORBmatcher(0.6, True)
    hbi, hbd = m.best_in_radius_host(probes, dup(s.mp_desc), s.kp, s.desc, s.bounds)
    assert np.array_equal(hbi, bi) and np.array_equal(hbd, bd)
    k = len(s.level)
    assert np.array_equal(hbi[:k], hbi[k:]) and (hbi[:k] >= 0).sum() > 200
    # no probes, and a keyframe without features
    e_bi, e_bd = m.best_in_radius_host(probes[:0], dup(s.mp_desc)[:0], s.kp, s.desc, s.bounds)
    assert len(e_bi) == 0 and len(e_bd) == 0
    z_bi, z_bd = m.best_in_radius_host(probes, dup(s.mp_desc), s.kp[:0], s.desc[:0], s.bounds)
    assert (z_bi == -1).all() and (z_bd == INT_MAX).all()


def two_views(orbref, rng, n, w, h, T1, T2, s12, pool=None):
    """n world points and two keyframes that see them: camera 1 at T1w, camera 2 at T2w in a world scaled by s12
    (p_c1 = s12 R12 p_c2 + t12).  Per keyframe: keypoints (jittered projections of the points in view, clutter,
    shuffled), descriptors, and per keypoint the index of its map point (-1 for clutter).  pool: the map points'
    descriptors are drawn from these rows (many near-equal descriptors) instead of being random."""
    K = (np.float32(520.0), np.float32(515.0), np.float32(w / 2 + 3.5), np.float32(h / 2 - 2.25))
    z = rng.uniform(2.0, 12.0, n)
    x = (rng.uniform(-40, w + 40, n) - K[2]) / K[0] * z
    y = (rng.uniform(-40, h + 40, n) - K[3]) / K[1] * z
    cam1 = np.stack([x, y, z], 1)
    R1, t1 = T1[:3, :3].astype(np.float64), T1[:3, 3].astype(np.float64)
    world = (cam1 - t1) @ R1  # R1^T (cam1 - t1)
    R2, t2 = T2[:3, :3].astype(np.float64), T2[:3, 3].astype(np.float64)
    cam2 = world @ R2.T + t2
    # camera 1 from camera 2: cam1 = R1 R2^T (cam2 - t2) + t1; the Sim3 carries the scale between the two maps
    R12 = R1 @ R2.T
    t12 = t1 - R12 @ t2
    cam2 = cam2 / s12
    T2s = T2.copy()
    T2s[:3, 3] = (t2 / s12).astype(np.float32)
    world2 = world / s12  # the points as keyframe 2's map holds them
    mp_desc = rng.integers(0, 256, (n, 32), dtype=np.uint8) if pool is None else pool[rng.integers(0, len(pool), n)].copy()
    level = rng.integers(0, 8, n)
    views = []
    for cam, pts_world in ((cam1, world), (cam2, world2)):
        u = K[0] * cam[:, 0] / cam[:, 2] + K[2] + rng.normal(0, 1.5, n)
        v = K[1] * cam[:, 1] / cam[:, 2] + K[3] + rng.normal(0, 1.5, n)
        seen = np.nonzero((u >= 0) & (u < w) & (v >= 0) & (v < h) & (cam[:, 2] > 0) & (rng.random(n) > 0.15))[0]
        nc = int(0.3 * n)
        m = len(seen) + nc
        kp = np.zeros(m, orbref.KP_DTYPE)
        kp["x"] = np.concatenate([u[seen], rng.uniform(0, w, nc)]).astype(np.float32)
        kp["y"] = np.concatenate([v[seen], rng.uniform(0, h, nc)]).astype(np.float32)
        kp["octave"] = np.clip(np.concatenate([level[seen], rng.integers(0, 8, nc)]) - rng.integers(0, 2, m), 0, 7)
        kp["angle"] = rng.uniform(0, 360, m).astype(np.float32)
        kp["size"], kp["class_id"] = 31.0, -1
        desc = np.concatenate([mp_desc[seen], rng.integers(0, 256, (nc, 32), dtype=np.uint8)])
        for i in range(m):
            for _ in range(rng.integers(0, 6)):
                desc[i, rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
        owner = np.concatenate([seen, np.full(nc, -1)])
        p = rng.permutation(m)
        dist = np.linalg.norm(cam, axis=1)
        views.append(dict(kp=kp[p], desc=desc[p], owner=owner[p], pos=pts_world.astype(np.float32), dist=dist))
    return K, views, T2s, R12.astype(np.float32), t12.astype(np.float32), mp_desc, level


@pytest.mark.parametrize("n,th", [(1800, 7.5), (0, 7.5), (1, 7.5), (60, 10.0)])
def test_search_by_sim3(orb, orbref, n, th):
    """th = 7.5 in LoopClosing::ComputeSim3 (src/LoopClosing.cc:347)."""
    rng = np.random.default_rng(191 + n)
    w, h, s12 = 752, 480, np.float32(1.07)
    T1, T2 = pose(rng), pose(rng)
    K, (v1, v2), T2s, R12, t12, mp_desc, level = two_views(orbref, rng, n, w, h, T1, T2, float(s12))
    bounds = [0, 0, w, h]
    side = []
    for v, other in ((v1, v2), (v2, v1)):
        k = len(v["kp"])
        has = v["owner"] >= 0
        o = np.where(has, v["owner"], 0)
        skip = (~has | (rng.random(k) < 0.1)).astype(np.uint8)  # NULL, bad or already matched
        pos = np.where(has[:, None], v["pos"][o] if n else np.zeros((k, 3), np.float32), 0).astype(np.float32)
        # the scale-invariance range is tied to the distance from the OTHER camera, where the point is searched
        d_other = other["dist"][o] if n else np.zeros(k)
        min_d = (d_other / SCALE[level[o] if n else np.zeros(k, int)] * rng.uniform(0.85, 1.18, k)).astype(np.float32)
        max_d = (min_d * rng.choice(np.array([1.1, 1.2 ** 7.5], np.float32), k, p=[0.1, 0.9])).astype(np.float32)
        d = np.where(has[:, None], mp_desc[o] if n else np.zeros((k, 32), np.uint8), 0).astype(np.uint8)
        side.append((skip, pos, min_d, max_d, d))
    ref = orbref.search_by_sim3(v1["kp"], v1["desc"], v2["kp"], v2["desc"], bounds, K, SCALE, T1, T2s, s12, R12, t12, th,
                                *side[0], *side[1])
    nfound, m12, vn1, vn2, pr1, pr2 = ref
    # probes of keyframe 1's map points search keyframe 2 and the other way round: one batch of two frames
    bi, bd, _ = device_best(orb, [(v2["kp"], v2["desc"]), (v1["kp"], v1["desc"])], bounds, [pr1, pr2], [side[0][4], side[1][4]])
    d1 = np.where(bd[0] <= TH_HIGH, bi[0], -1)
    d2 = np.where(bd[1] <= TH_HIGH, bi[1], -1)
    assert np.array_equal(d1, vn1) and np.array_equal(d2, vn2), "vnMatch1 / vnMatch2 differ from the oracle's"
    agree = np.full(len(d1), -1, np.int32)
    for i1, idx2 in enumerate(d1):
        if idx2 >= 0 and d2[idx2] == i1:
            agree[i1] = idx2
    assert np.array_equal(agree, m12) and int((agree >= 0).sum()) == nfound, "agreement differs from the oracle's"
    if n >= 1000:
        # the pairs that agree are keypoints of the same world point
        ok = agree >= 0
        assert nfound > 300 and (v1["owner"][ok] == v2["owner"][agree[ok]]).mean() > 0.95


def test_best_in_radius_argument_errors(orb):
    import torch
    m = orb.ORBmatcher(0.6, True)
    kp = np.zeros(4, orb.KP_DTYPE)
    F = orb.FrameBatch.from_host(m, [kp], [np.zeros((4, 32), np.uint8)], [0, 0, 640, 480])
    z = torch.zeros((1, 2, 32), dtype=torch.uint8, device="cuda")
    n = torch.zeros(1, dtype=torch.int32, device="cuda")
    bi, bd = m.BestInRadius(F, z, z, n)
    m.synchronize()
    assert (bi.cpu().numpy() == -1).all() and (bd.cpu().numpy() == INT_MAX).all()
    L = orb._lib.lib()
    with pytest.raises(orb.OrbhipError):
        orb._lib.check(L.orbhip_best_in_radius_device(m._h, 0, z.data_ptr(), n.data_ptr(), 2, F.kp.data_ptr(), F.desc.data_ptr(),
                                                      F.cell_start.data_ptr(), F.cell_items.data_ptr(), 1, F.cap, F.bounds,
                                                      z.data_ptr(), z.data_ptr()))
    with pytest.raises(orb.OrbhipError):
        orb._lib.check(L.orbhip_best_in_radius_device(m._h, z.data_ptr(), z.data_ptr(), n.data_ptr(), 0, F.kp.data_ptr(),
                                                      F.desc.data_ptr(), F.cell_start.data_ptr(), F.cell_items.data_ptr(), 1, F.cap,
                                                      F.bounds, z.data_ptr(), z.data_ptr()))
