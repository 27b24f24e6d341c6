"""Parity of the device search / assignment loops of the five ORBmatcher::SearchByProjection overloads
(src/ORBmatcher.cc:52-128, :297-418, :530-605, :1518-1631, :1633-1757) against the oracle's literal
restatements (oracle/frame_ref.cpp), through the C ABI.  The oracle projects the map points as the
reference does and returns the probe it built per map point; the device receives exactly those probes, so
the comparison covers GetFeaturesInArea, the skip of features that already hold a map point, the Hamming
distances, the best / second-best rules with their level test, and the rotation histogram.  Bar: bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SCALE = (1.2 ** np.arange(8)).astype(np.float32)


def rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def pose(rng, scale=1.0, big=False):
    a = 0.3 if big else 0.03
    T = np.eye(4)
    T[:3, :3] = scale * rot(*rng.uniform(-a, a, 3))
    T[:3, 3] = scale * rng.uniform(-0.2, 0.2, 3)
    return T.astype(np.float32)


class Scene:
    """n world points seen by a pinhole camera at pose T: the frame's keypoints are their jittered projections
    (plus clutter), descriptors are perturbed copies of the points' descriptors."""

    def __init__(self, orbref, rng, n, w, h, T, clutter=0.3, ties=False):
        self.w, self.h = w, h
        self.K = (np.float32(520.0), np.float32(515.0), np.float32(w / 2 + 3.5), np.float32(h / 2 - 2.25))
        z = rng.uniform(2.0, 12.0, n)
        x = (rng.uniform(-40, w + 40, n) - self.K[2]) / self.K[0] * z
        y = (rng.uniform(-40, h + 40, n) - self.K[3]) / self.K[1] * z
        cam = np.stack([x, y, z], 1)
        R, t = T[:3, :3].astype(np.float64), T[:3, 3].astype(np.float64)
        s = np.cbrt(np.linalg.det(R))
        self.pos = ((cam - t) @ (R / s) / s).astype(np.float32)  # world points: T maps them to `cam`
        pool = rng.integers(0, 256, (10, 32), dtype=np.uint8)
        self.mp_desc = pool[rng.integers(0, 10, n)].copy() if ties else rng.integers(0, 256, (n, 32), dtype=np.uint8)
        if ties:
            for i in range(n):
                for _ in range(rng.integers(0, 3)):
                    self.mp_desc[i, rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
        self.level = rng.integers(0, 8, n).astype(np.int32)
        self.angle = rng.uniform(0, 360, n).astype(np.float32)
        # the frame
        u = self.K[0] * x / z + self.K[2] + rng.normal(0, 2.0, n)
        v = self.K[1] * y / z + self.K[3] + rng.normal(0, 2.0, n)
        nc = int(clutter * n)
        m = n + nc
        kp = np.zeros(m, orbref.KP_DTYPE)
        kp["x"] = np.concatenate([u, rng.uniform(0, w, nc)]).astype(np.float32)
        kp["y"] = np.concatenate([v, rng.uniform(0, h, nc)]).astype(np.float32)
        kp["octave"] = np.clip(np.concatenate([self.level, rng.integers(0, 8, nc)]) + rng.integers(-1, 2, m), 0, 7)
        kp["angle"] = np.mod(np.concatenate([self.angle, rng.uniform(0, 360, nc)]) + 25.0 + rng.normal(0, 6, m), 360).astype(np.float32)
        wild = rng.random(m) < 0.1
        kp["angle"][wild] = rng.uniform(0, 360, wild.sum()).astype(np.float32)
        kp["size"], kp["class_id"] = 31.0, -1
        desc = np.concatenate([self.mp_desc, rng.integers(0, 256, (nc, 32), dtype=np.uint8)])
        for i in range(m):
            for _ in range(rng.integers(0, 5)):
                desc[i, rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
        keep = rng.random(m) > 0.1
        p = rng.permutation(int(keep.sum()))
        self.kp, self.desc = kp[keep][p], desc[keep][p]
        self.taken = (rng.random(len(self.kp)) < 0.15).astype(np.uint8)
        self.bounds = [0, 0, w, h]


def run_device(orb, scenes, results, rule, ratio=0.9, th_dist=100, ori=False):
    """scenes: target frames; results: the oracle's (nmatches, match_of_feature, probes, probe_desc, taken) per frame."""
    import torch
    m = orb.ORBmatcher(ratio, ori)
    F = orb.FrameBatch.from_host(m, [s.kp for s in scenes], [s.desc for s in scenes], scenes[0].bounds)
    B, pcap = len(scenes), max(max(len(r[2]) for r in results), 1)
    probes = np.zeros((B, pcap), orb.PROBE_DTYPE)
    pdesc = np.zeros((B, pcap, 32), np.uint8)
    taken = np.zeros((B, F.cap), np.uint8)
    npr = np.zeros(B, np.int32)
    for f, r in enumerate(results):
        k = len(r[2])
        npr[f] = k
        probes[f, :k] = r[2]
        pdesc[f, :k] = r[3]
        taken[f, :len(r[4])] = r[4]
    probes[:, :]["reserved"] = 0
    d_probes = torch.from_numpy(probes.view(np.uint8).reshape(B, pcap, 32).copy()).cuda()
    nm, mo = m.SearchByProjection(F, d_probes, torch.from_numpy(pdesc).cuda(), torch.from_numpy(npr).cuda(), rule,
                                  taken=torch.from_numpy(taken).cuda(), th_dist=th_dist)
    nm, mo = nm.cpu().numpy(), mo.cpu().numpy()
    total = 0
    for f, r in enumerate(results):
        n = len(scenes[f].kp)
        assert np.array_equal(r[1], mo[f, :n]), "frame %d: %d assignments differ" % (f, int((r[1] != mo[f, :n]).sum()))
        assert (mo[f, n:] == -1).all()
        assert r[0] == nm[f], "frame %d nmatches" % f
        assert r[0] == int((r[1] >= 0).sum())
        total += r[0]
    return total


SIZES = [1500, 0, 1, 70, 3000]


@pytest.mark.parametrize("th", [1.0, 3.0])
@pytest.mark.parametrize("ties", [False, True])
def test_local_map_overload(orb, orbref, th, ties):
    rng = np.random.default_rng(101 + ties)
    scenes, results = [], []
    for n in SIZES:
        T = pose(rng)
        s = Scene(orbref, rng, n, 752, 480, T, ties=ties)
        # Frame::isInFrustum's outputs (src/Frame.cc:191-196), synthesised: projection, level, viewing cosine
        cam = s.pos.astype(np.float64) @ T[:3, :3].T.astype(np.float64) + T[:3, 3]
        px = (s.K[0] * cam[:, 0] / cam[:, 2] + s.K[2]).astype(np.float32)
        py = (s.K[1] * cam[:, 1] / cam[:, 2] + s.K[3]).astype(np.float32)
        in_view = (rng.random(n) < 0.9).astype(np.uint8)
        bad = (rng.random(n) < 0.05).astype(np.uint8)
        cosv = rng.choice(np.array([0.9, 0.998, 0.9981, 0.9995, 1.0], np.float32), n)
        r = orbref.sbp_local_map(s.kp, s.desc, s.bounds, SCALE, s.taken, in_view, bad, s.level, cosv, px, py, s.mp_desc, th, 0.8)
        scenes.append(s)
        results.append(r + (s.mp_desc, s.taken))
    assert run_device(orb, scenes, results, orb.ORBmatcher.PROJ_LOCAL_MAP, ratio=0.8) > 1000


@pytest.mark.parametrize("ties", [False, True])
def test_sim3_overload(orb, orbref, ties):
    rng = np.random.default_rng(111 + ties)
    scenes, results = [], []
    for n in SIZES:
        S = pose(rng, scale=1.3, big=False)
        s = Scene(orbref, rng, n, 640, 480, S, ties=ties)
        R = S[:3, :3].astype(np.float64) / 1.3
        Ow = -R.T @ (S[:3, 3].astype(np.float64) / 1.3)
        PO = s.pos.astype(np.float64) - Ow
        dist = np.linalg.norm(PO, axis=1)
        normal = (PO / np.maximum(dist[:, None], 1e-9) + rng.normal(0, 0.5, (n, 3))).astype(np.float32)
        min_d = (dist / 1.2 ** rng.uniform(-1, 8, n)).astype(np.float32)
        max_d = (min_d * 1.2 ** 7.5).astype(np.float32)
        skip = (rng.random(n) < 0.1).astype(np.uint8)
        r = orbref.sbp_sim3(s.kp, s.desc, s.bounds, s.K, SCALE, S, s.taken, skip, s.pos, normal, min_d, max_d, s.mp_desc, 10)
        scenes.append(s)
        results.append(r + (s.mp_desc, s.taken))
    assert run_device(orb, scenes, results, orb.ORBmatcher.PROJ_BEST, th_dist=20) > 300


@pytest.mark.parametrize("window,ratio", [(15, 0.8), (60, 0.95)])
def test_frame_to_frame_overload(orb, orbref, window, ratio):
    rng = np.random.default_rng(121)
    scenes, results = [], []
    for n in SIZES:
        T2 = pose(rng)
        s2 = Scene(orbref, rng, n, 640, 480, T2, ties=True)
        # F1: one keypoint per world point (octave = the point's level), descriptors of the points
        kp1 = np.zeros(n, orbref.KP_DTYPE)
        kp1["octave"] = s2.level
        skip1 = (rng.random(n) < 0.2).astype(np.uint8)
        r = orbref.sbp_frames(kp1, s2.mp_desc, skip1, s2.pos, T2, s2.K, s2.kp, s2.desc, s2.bounds, s2.taken, window, ratio)
        scenes.append(s2)
        results.append(r + (s2.mp_desc, s2.taken))
    assert run_device(orb, scenes, results, orb.ORBmatcher.PROJ_RATIO, ratio=ratio) > 300


@pytest.mark.parametrize("th,ori", [(15.0, True), (7.0, True), (15.0, False)])
def test_last_frame_overload(orb, orbref, th, ori):
    rng = np.random.default_rng(131)
    scenes, results = [], []
    for n in SIZES:
        T = pose(rng)
        s = Scene(orbref, rng, n, 1280, 720, T, ties=True)
        kp_last = np.zeros(n, orbref.KP_DTYPE)
        kp_last["octave"], kp_last["angle"] = s.level, s.angle
        skip = (rng.random(n) < 0.2).astype(np.uint8)
        r = orbref.sbp_last_frame(s.kp, s.desc, s.bounds, s.taken, s.K, T, SCALE, kp_last, s.mp_desc, skip, s.pos, th, ori)
        scenes.append(s)
        results.append(r + (s.mp_desc, s.taken))
    assert run_device(orb, scenes, results, orb.ORBmatcher.PROJ_BEST, th_dist=100, ori=ori) > 500


@pytest.mark.parametrize("orb_dist,ori", [(100, True), (64, False)])
def test_keyframe_overload(orb, orbref, orb_dist, ori):
    rng = np.random.default_rng(141)
    scenes, results = [], []
    for n in SIZES:
        T = pose(rng)
        s = Scene(orbref, rng, n, 752, 480, T, ties=True)
        R = T[:3, :3].astype(np.float64)
        Ow = -R.T @ T[:3, 3].astype(np.float64)
        dist = np.linalg.norm(s.pos.astype(np.float64) - Ow, axis=1)
        min_d = (dist / SCALE[s.level] * rng.uniform(0.95, 1.05, n)).astype(np.float32)
        skip = (rng.random(n) < 0.2).astype(np.uint8)
        r = orbref.sbp_keyframe(s.kp, s.desc, s.bounds, s.taken, s.K, T, SCALE, skip, s.pos, min_d, s.mp_desc, s.angle, 10.0,
                                orb_dist, ori)
        scenes.append(s)
        results.append(r + (s.mp_desc, s.taken))
    assert run_device(orb, scenes, results, orb.ORBmatcher.PROJ_BEST, th_dist=orb_dist, ori=ori) > 500


def test_probe_argument_errors(orb):
    import torch
    m = orb.ORBmatcher(0.9, True)
    kp = np.zeros(4, orb.KP_DTYPE)
    F = orb.FrameBatch.from_host(m, [kp], [np.zeros((4, 32), np.uint8)], [0, 0, 640, 480])
    z = torch.zeros((1, 2, 32), dtype=torch.uint8, device="cuda")
    n = torch.zeros(1, dtype=torch.int32, device="cuda")
    with pytest.raises(orb.OrbhipError):
        m.SearchByProjection(F, z, z, n, 7)
    nm, mo = m.SearchByProjection(F, z, z, n, orb.ORBmatcher.PROJ_BEST)
    assert int(nm[0]) == 0 and (mo.cpu().numpy() == -1).all()


def test_probe_lists_longer_than_the_lds_run_buffer(orb, orbref):
    """One level, a 200x150 frame and a radius that covers it: every probe's candidate list holds all of the
    frame's ~5000 features (more than the 4096 entries staged in LDS at a time)."""
    rng = np.random.default_rng(151)
    T = pose(rng)
    s = Scene(orbref, rng, 4200, 200, 150, T, ties=True)
    s.kp["octave"] = 1
    s.level[:] = 1
    n = len(s.level)
    kp_last = np.zeros(n, orbref.KP_DTYPE)
    kp_last["octave"], kp_last["angle"] = s.level, s.angle
    skip = (rng.random(n) < 0.1).astype(np.uint8)
    r = orbref.sbp_last_frame(s.kp, s.desc, s.bounds, s.taken, s.K, T, SCALE, kp_last, s.mp_desc, skip, s.pos, 300.0, True)
    assert len(s.kp) > 4096
    assert run_device(orb, [s], [r + (s.mp_desc, s.taken)], orb.ORBmatcher.PROJ_BEST, th_dist=100, ori=True) > 500
