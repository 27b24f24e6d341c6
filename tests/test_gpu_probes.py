"""The device probe builder (orbhip_build_probes_device) against the probes the oracle's restatements of the
projection matchers build (oracle/frame_ref.cpp): pose arithmetic as OpenCV 2.4 evaluates it on CV_32F matrices,
depth / image / distance-range / viewing-angle tests, predicted level, radius — src/ORBmatcher.cc:1531-1556,
:1651-1690, :63-81 with Frame::isInFrustum (src/Frame.cc:138-199), :1054-1094, :1180-1223, :326-375.  Bar: every
probe record equal byte for byte; and the matcher fed with device-built probes returns the oracle's matches."""
import numpy as np
import pytest

from test_gpu_fuse import map_point_geometry
from test_gpu_projection import SCALE, Scene, pose

pytestmark = pytest.mark.gpu

SIZES = [1500, 0, 1, 70, 3000]


def batch_of(arrs, pcap, shape, dtype):
    out = np.zeros((len(arrs), pcap) + shape, dtype)
    for f, a in enumerate(arrs):
        out[f, :len(a)] = a
    return out


def device_probes(orb, kind, scenes, poses, th, sim3=False, normal=None, min_d=None, max_d=None, src_kp=None, skip=None, limit=0.5):
    import torch
    m = orb.ORBmatcher(0.8, True)
    pcap = max(max(len(s.level) for s in scenes), 1)
    cu = lambda a: None if a is None else torch.from_numpy(a).cuda()  # noqa: E731
    cams = m.projections(poses, scenes[0].K, sim3=sim3)
    pos = batch_of([s.pos for s in scenes], pcap, (3,), np.float32)
    npts = np.array([len(s.level) for s in scenes], np.int32)
    kp = None if src_kp is None else batch_of([k.view(np.uint8).reshape(-1, 28) for k in src_kp], pcap, (28,), np.uint8)
    probes = m.BuildProbes(kind, cams, scenes[0].bounds, SCALE, cu(pos), cu(npts), th,
                           normal=None if normal is None else cu(batch_of(normal, pcap, (3,), np.float32)),
                           min_dist=None if min_d is None else cu(batch_of(min_d, pcap, (), np.float32)),
                           max_dist=None if max_d is None else cu(batch_of(max_d, pcap, (), np.float32)),
                           src_kp=cu(kp), skip=None if skip is None else cu(batch_of(skip, pcap, (), np.uint8)),
                           viewing_cos_limit=limit)
    m.synchronize()
    return m, probes


def check(orb, probes, refs):
    got = probes.cpu().numpy()
    valid = 0
    for f, r in enumerate(refs):
        want = np.ascontiguousarray(r).view(np.uint8).reshape(-1, 32)
        assert np.array_equal(got[f, :len(want)], want), "frame %d: %d probes differ" % (
            f, int((got[f, :len(want)] != want).any(axis=1).sum()))
        assert not got[f, len(want):].any(), "frame %d: slots past the frame's points" % f
        valid += int(r["valid"].sum())
    return valid


@pytest.mark.parametrize("th", [15.0, 7.0])
def test_last_frame_probes(orb, orbref, th):
    rng = np.random.default_rng(201)
    scenes, poses, refs, kps, skips = [], [], [], [], []
    for n in SIZES:
        T = pose(rng)
        s = Scene(orbref, rng, n, 1280, 720, T)
        kp_last = np.zeros(n, orbref.KP_DTYPE)
        kp_last["octave"], kp_last["angle"] = s.level, s.angle
        skip = (rng.random(n) < 0.2).astype(np.uint8)
        refs.append(orbref.sbp_last_frame(s.kp, s.desc, s.bounds, s.taken, s.K, T, SCALE, kp_last, s.mp_desc, skip, s.pos, th, True)[2])
        scenes.append(s), poses.append(T), kps.append(kp_last), skips.append(skip)
    _, probes = device_probes(orb, orb.ORBmatcher.PROBES_LAST_FRAME, scenes, poses, th, src_kp=kps, skip=skips)
    assert check(orb, probes, refs) > 2000


def test_keyframe_probes(orb, orbref):
    rng = np.random.default_rng(211)
    scenes, poses, refs, kps, skips, mins = [], [], [], [], [], []
    for n in SIZES:
        T = pose(rng, big=True)
        s = Scene(orbref, rng, n, 752, 480, T)
        _, min_d, _ = map_point_geometry(rng, s, T)
        kp_kf = np.zeros(n, orbref.KP_DTYPE)
        kp_kf["angle"] = s.angle
        skip = (rng.random(n) < 0.2).astype(np.uint8)
        refs.append(orbref.sbp_keyframe(s.kp, s.desc, s.bounds, s.taken, s.K, T, SCALE, skip, s.pos, min_d, s.mp_desc, s.angle, 10.0,
                                        100, True)[2])
        scenes.append(s), poses.append(T), kps.append(kp_kf), skips.append(skip), mins.append(min_d)
    _, probes = device_probes(orb, orb.ORBmatcher.PROBES_KEYFRAME, scenes, poses, 10.0, min_d=mins, src_kp=kps, skip=skips)
    assert check(orb, probes, refs) > 2000


@pytest.mark.parametrize("th,limit", [(3.0, 0.5), (1.0, 0.5), (5.0, 0.9)])
def test_frustum_probes_and_the_local_map_search(orb, orbref, th, limit):
    """Frame::isInFrustum + SearchByProjection(F, vpMapPoints, th) as Tracking::SearchLocalPoints runs them
    (src/Tracking.cc: th = 1, or 5 after a relocalisation; the map's th = 3 default)."""
    import torch
    rng = np.random.default_rng(221)
    scenes, poses, refs, results, normals, mins, maxs, bads = [], [], [], [], [], [], [], []
    for n in SIZES:
        T = pose(rng)
        s = Scene(orbref, rng, n, 752, 480, T, ties=True)
        normal, min_d, max_d = map_point_geometry(rng, s, T)
        # view cosines around RadiusByViewingCos's 0.998: some normals point straight at the camera
        R = T[:3, :3].astype(np.float64)
        Ow = -R.T @ T[:3, 3].astype(np.float64)
        PO = s.pos.astype(np.float64) - Ow
        straight = rng.random(n) < 0.4
        normal[straight] = (PO / np.maximum(np.linalg.norm(PO, axis=1)[:, None], 1e-9)).astype(np.float32)[straight]
        bad = (rng.random(n) < 0.05).astype(np.uint8)
        in_view, px, py, lv, vc = orbref.is_in_frustum(s.bounds, s.K, T, SCALE, s.pos, normal, min_d, max_d, limit)
        r = orbref.sbp_local_map(s.kp, s.desc, s.bounds, SCALE, s.taken, in_view, bad, lv, vc, px, py, s.mp_desc, th, 0.8)
        refs.append(r[2]), results.append(r)
        scenes.append(s), poses.append(T), normals.append(normal), mins.append(min_d), maxs.append(max_d), bads.append(bad)
    m, probes = device_probes(orb, orb.ORBmatcher.PROBES_FRUSTUM, scenes, poses, th, normal=normals, min_d=mins, max_d=maxs,
                              skip=bads, limit=limit)
    nvalid = check(orb, probes, refs)
    assert nvalid > 1500
    radii = np.concatenate([r["r"][r["valid"] != 0] / SCALE[r["max_level"][r["valid"] != 0]] for r in refs])
    assert len(np.unique(np.round(radii / (th if th != 1.0 else 1.0), 3))) == 2  # both radii of RadiusByViewingCos occur
    # the device chain: probes -> search, against the oracle's matches
    F = orb.FrameBatch.from_host(m, [s.kp for s in scenes], [s.desc for s in scenes], scenes[0].bounds)
    pcap = probes.shape[1]
    pdesc = torch.from_numpy(batch_of([s.mp_desc for s in scenes], pcap, (32,), np.uint8)).cuda()
    npts = torch.from_numpy(np.array([len(s.level) for s in scenes], np.int32)).cuda()
    taken = torch.from_numpy(batch_of([s.taken for s in scenes], F.cap, (), np.uint8)).cuda()
    nm, mo = m.SearchByProjection(F, probes, pdesc, npts, orb.ORBmatcher.PROJ_LOCAL_MAP, taken=taken, check_orientation=False)
    nm, mo = nm.cpu().numpy(), mo.cpu().numpy()
    for f, r in enumerate(results):
        assert r[0] == nm[f] and np.array_equal(r[1], mo[f, :len(r[1])]), "frame %d: matches" % f


@pytest.mark.parametrize("form", ["pose", "sim3", "sim3_search"])
def test_fuse_probes(orb, orbref, form):
    rng = np.random.default_rng(231)
    sim3 = form != "pose"
    scale = 1.3 if sim3 else 1.0
    scenes, poses, refs, normals, mins, maxs, skips = [], [], [], [], [], [], []
    th = 10 if form == "sim3_search" else 3.0
    for n in SIZES:
        T = pose(rng, scale=scale)
        s = Scene(orbref, rng, n, 640, 480, T)
        normal, min_d, max_d = map_point_geometry(rng, s, T, scale)
        skip = (rng.random(n) < 0.1).astype(np.uint8)
        if form == "sim3_search":
            refs.append(orbref.sbp_sim3(s.kp, s.desc, s.bounds, s.K, SCALE, T, s.taken, skip, s.pos, normal, min_d, max_d, s.mp_desc, th)[2])
        else:
            refs.append(orbref.fuse(s.kp, s.desc, s.bounds, s.K, SCALE, T, skip, s.pos, normal, min_d, max_d, s.mp_desc, th, sim3=sim3)[3])
        scenes.append(s), poses.append(T), normals.append(normal), mins.append(min_d), maxs.append(max_d), skips.append(skip)
    _, probes = device_probes(orb, orb.ORBmatcher.PROBES_FUSE, scenes, poses, th, sim3=sim3, normal=normals, min_d=mins, max_d=maxs,
                              skip=skips)
    assert check(orb, probes, refs) > 1500


@pytest.mark.parametrize("window", [15, 50])
def test_window_probes(orb, orbref, window):
    """SearchByProjection(F1, F2, windowSize, ...) (src/ORBmatcher.cc:545-566): radius = the window, the source
    keypoint's octave the only level, no depth or image test."""
    rng = np.random.default_rng(241)
    scenes, poses, refs, kps, skips = [], [], [], [], []
    for n in SIZES:
        T2 = pose(rng)
        s2 = Scene(orbref, rng, n, 640, 480, T2)
        kp1 = np.zeros(n, orbref.KP_DTYPE)
        kp1["octave"] = s2.level
        skip1 = (rng.random(n) < 0.2).astype(np.uint8)
        refs.append(orbref.sbp_frames(kp1, s2.mp_desc, skip1, s2.pos, T2, s2.K, s2.kp, s2.desc, s2.bounds, s2.taken, window, 0.8)[2])
        scenes.append(s2), poses.append(T2), kps.append(kp1), skips.append(skip1)
    _, probes = device_probes(orb, orb.ORBmatcher.PROBES_WINDOW, scenes, poses, float(window), src_kp=kps, skip=skips)
    assert check(orb, probes, refs) > 2000


@pytest.mark.parametrize("n,th", [(1800, 7.5), (0, 7.5), (1, 7.5), (60, 10.0)])
def test_sim3_probes(orb, orbref, n, th):
    """Both directions of SearchBySim3 (src/ORBmatcher.cc:1330-1375, :1415-1460): pose of the point's own keyframe, then
    the similarity, each product rounded to float; distance range on the distance in the target camera."""
    import torch
    from test_gpu_fuse import two_views
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
        skip = (~has | (rng.random(k) < 0.1)).astype(np.uint8)
        pos = np.where(has[:, None], v["pos"][o] if n else np.zeros((k, 3), np.float32), 0).astype(np.float32)
        d_other = other["dist"][o] if n else np.zeros(k)
        min_d = (d_other / SCALE[level[o] if n else np.zeros(k, int)] * rng.uniform(0.85, 1.18, k)).astype(np.float32)
        max_d = (min_d * rng.choice(np.array([1.1, 1.2 ** 7.5], np.float32), k, p=[0.1, 0.9])).astype(np.float32)
        d = np.where(has[:, None], mp_desc[o] if n else np.zeros((k, 32), np.uint8), 0).astype(np.uint8)
        side.append((skip, pos, min_d, max_d, d))
    ref = orbref.search_by_sim3(v1["kp"], v1["desc"], v2["kp"], v2["desc"], bounds, K, SCALE, T1, T2s, s12, R12, t12, th,
                                *side[0], *side[1])
    m = orb.ORBmatcher(0.75, True)
    to2, to1 = m.sim3_projections(s12, R12, t12, K)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    valid = 0
    for (skip, pos, min_d, max_d, _), T, to, want in ((side[0], T1, to2, ref[4]), (side[1], T2s, to1, ref[5])):
        k = len(skip)
        pcap = max(k, 1)
        pad = lambda a, shape: np.concatenate([a, np.zeros((pcap - k,) + shape, a.dtype)]) if k < pcap else a  # noqa: E731
        probes = m.BuildProbesSim3(m.projections([T], K), to, bounds, SCALE, cu(pad(pos, (3,))[None]), cu(np.array([k], np.int32)), th,
                                   cu(pad(min_d, ())[None]), cu(pad(max_d, ())[None]), skip=cu(pad(skip, ())[None]))
        m.synchronize()
        got = probes.cpu().numpy()[0, :k]
        exp = np.ascontiguousarray(want).view(np.uint8).reshape(-1, 32)
        assert np.array_equal(got, exp), "%d probes differ" % int((got != exp).any(axis=1).sum())
        valid += int(want["valid"].sum())
    if n >= 1000:
        assert valid > 1000


def test_build_probes_argument_errors(orb):
    import torch
    m = orb.ORBmatcher(0.8, True)
    cams = m.projections([np.eye(4, dtype=np.float32)], (500.0, 500.0, 320.0, 240.0))
    pos = torch.zeros((1, 4, 3), dtype=torch.float32, device="cuda")
    n = torch.zeros(1, dtype=torch.int32, device="cuda")
    p = m.BuildProbes(orb.ORBmatcher.PROBES_LAST_FRAME, cams, [0, 0, 640, 480], SCALE, pos, n, 15.0,
                      src_kp=torch.zeros((1, 4, 28), dtype=torch.uint8, device="cuda"))
    m.synchronize()
    assert not p.cpu().numpy().any()
    with pytest.raises(orb.OrbhipError):  # the keypoints that carry octave and angle are missing
        m.BuildProbes(orb.ORBmatcher.PROBES_LAST_FRAME, cams, [0, 0, 640, 480], SCALE, pos, n, 15.0)
    with pytest.raises(orb.OrbhipError):  # normals and distance ranges are missing
        m.BuildProbes(orb.ORBmatcher.PROBES_FUSE, cams, [0, 0, 640, 480], SCALE, pos, n, 3.0)
    with pytest.raises(orb.OrbhipError):
        m.BuildProbes(7, cams, [0, 0, 640, 480], SCALE, pos, n, 3.0)
