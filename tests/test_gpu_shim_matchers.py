"""The forwarding C++ bodies of orb-slam2_amd/shim/ORBmatcher.cc for the matcher methods that work on
poses, map points and BoW vectors: four ORBmatcher::SearchByProjection overloads (src/ORBmatcher.cc:297-418,
:530-605, :1518-1631, :1633-1757), both ORBmatcher::SearchByBoW (:158-295, :726-861),
MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:191-256), Frame::UndistortKeyPoints and
Frame::ComputeBoW (src/Frame.cc:281-320).  shim_test builds stand-ins of Frame / KeyFrame / MapPoint from
the blobs written here, calls the body as the reference's threads would (pose arithmetic on cv::Mat included),
and the results are compared with the oracle on the same inputs.  Bar: bit-exact."""
import os
import subprocess

import numpy as np
import pytest

from test_gpu_bow import _features, _frames, _second_view
from test_gpu_projection import SCALE, Scene, pose

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "..", "orb-slam2_amd", "shim", "shim_test")


def run(tmp_path, mode, arrays):
    assert os.path.exists(EXE), "build it with make -C orb-slam2_amd/shim (done by __graft_entry__.build)"
    fin, fout = tmp_path / (mode + ".in"), tmp_path / (mode + ".out")
    with open(fin, "wb") as f:
        for a in arrays:
            b = np.ascontiguousarray(a).tobytes()
            f.write(np.int32(len(b)).tobytes())
            f.write(b)
    r = subprocess.run([EXE, "--matchers", mode, str(fin), str(fout)], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, (mode, r.returncode, r.stdout, r.stderr)
    raw = fout.read_bytes()
    out, o = [], 0
    while o < len(raw):
        nb = int(np.frombuffer(raw[o:o + 4], np.int32)[0])
        out.append(raw[o + 4:o + 4 + nb])
        o += 4 + nb
    return out


def kinds_for(rng, n, ntaken, with_null=True):
    """Per map point: 0 usable, 1 NULL pointer, 2 isBad(), 3 already found / outlier (never more of those than
    features that hold a map point on entry, because that is where the test program puts them)."""
    k = rng.choice(np.array([0, 1, 2, 3], np.uint8), n, p=[0.8, 0.07 if with_null else 0.0, 0.08 if with_null else 0.15, 0.05])
    three = np.nonzero(k == 3)[0]
    k[three[ntaken:]] = 0
    return k


def projection_case(tmp_path, orbref, mode, s, T, kinds, ip, fp, normal, min_d, max_d, kp_src):
    n = len(s.level)
    z3 = np.zeros((n, 3), np.float32)
    blobs = [np.array(ip, np.int32), np.array(fp, np.float32), np.array(s.bounds, np.int32), s.kp, s.desc, s.taken,
             np.asarray(T, np.float32), SCALE, kinds, s.pos, z3 if normal is None else normal,
             np.zeros(n, np.float32) if min_d is None else min_d, np.zeros(n, np.float32) if max_d is None else max_d,
             s.mp_desc, kp_src]
    out = run(tmp_path, mode, blobs)
    return int(np.frombuffer(out[0], np.int32)[0]), np.frombuffer(out[1], np.int32)


@pytest.mark.parametrize("n", [1200, 0, 1])
def test_projection_overloads_through_the_cpp_bodies(orb, orbref, tmp_path, n):
    rng = np.random.default_rng(500 + n)
    total = 0
    # --- (KeyFrame*, Scw, vpPoints, vpMatched, th) ---
    S = pose(rng, scale=1.3)
    s = Scene(orbref, rng, n, 640, 480, S, ties=True)
    R = S[:3, :3].astype(np.float64) / 1.3
    Ow = -R.T @ (S[:3, 3].astype(np.float64) / 1.3)
    PO = s.pos.astype(np.float64) - Ow
    dist = np.linalg.norm(PO, axis=1)
    normal = (PO / np.maximum(dist[:, None], 1e-9) + rng.normal(0, 0.5, (n, 3))).astype(np.float32)
    min_d = (dist / 1.2 ** rng.uniform(-1, 8, n)).astype(np.float32)
    max_d = (min_d * 1.2 ** 7.5).astype(np.float32)
    kinds = kinds_for(rng, n, int(s.taken.sum()), with_null=False)
    rn, rmatch, _ = orbref.sbp_sim3(s.kp, s.desc, s.bounds, s.K, SCALE, S, s.taken, (kinds != 0).astype(np.uint8), s.pos, normal,
                                    min_d, max_d, s.mp_desc, 10)
    nm, owner = projection_case(tmp_path, orbref, "sim3", s, S, kinds, [10, 0, 0, 0], list(s.K) + [0.9, 0.0], normal, min_d,
                                max_d, np.zeros(n, orbref.KP_DTYPE))
    assert nm == rn and np.array_equal(owner, np.where(s.taken != 0, -2, rmatch)), "sim3 overload"
    total += rn
    # --- (Frame &F1, Frame &F2, windowSize, vpMapPointMatches2) ---
    T = pose(rng)
    s = Scene(orbref, rng, n, 640, 480, T, ties=True)
    kp1 = np.zeros(n, orbref.KP_DTYPE)
    kp1["octave"] = s.level
    kinds = kinds_for(rng, n, int(s.taken.sum()))
    rn, rmatch, _ = orbref.sbp_frames(kp1, s.mp_desc, (kinds != 0).astype(np.uint8), s.pos, T, s.K, s.kp, s.desc, s.bounds,
                                      s.taken, 30, np.float32(0.9))
    nm, owner = projection_case(tmp_path, orbref, "frames", s, T, kinds, [0, 30, 0, 0], list(s.K) + [0.9, 0.0], None, None, None,
                                kp1)
    assert nm == rn and np.array_equal(owner, np.where(s.taken != 0, -2, rmatch)), "frame-to-frame overload"
    total += rn
    # --- (Frame &CurrentFrame, const Frame &LastFrame, th) ---
    T = pose(rng)
    s = Scene(orbref, rng, n, 1280, 720, T, ties=True)
    kp_last = np.zeros(n, orbref.KP_DTYPE)
    kp_last["octave"], kp_last["angle"] = s.level, s.angle
    kinds = kinds_for(rng, n, int(s.taken.sum()))
    rn, rmatch, _ = orbref.sbp_last_frame(s.kp, s.desc, s.bounds, s.taken, s.K, T, SCALE, kp_last, s.mp_desc,
                                          (kinds != 0).astype(np.uint8), s.pos, 15.0, True)
    nm, owner = projection_case(tmp_path, orbref, "last", s, T, kinds, [0, 0, 0, 1], list(s.K) + [0.9, 15.0], None, None, None,
                                kp_last)
    assert nm == rn and np.array_equal(owner, np.where(s.taken != 0, -2, rmatch)), "last-frame overload"
    total += rn
    # --- (Frame &CurrentFrame, KeyFrame*, sAlreadyFound, th, ORBdist) ---
    T = pose(rng)
    s = Scene(orbref, rng, n, 752, 480, T, ties=True)
    R = T[:3, :3].astype(np.float64)
    Ow = -R.T @ T[:3, 3].astype(np.float64)
    dist = np.linalg.norm(s.pos.astype(np.float64) - Ow, axis=1)
    min_d = (dist / SCALE[s.level] * rng.uniform(0.95, 1.05, n)).astype(np.float32)
    kp_kf = np.zeros(n, orbref.KP_DTYPE)
    kp_kf["angle"] = s.angle
    kinds = kinds_for(rng, n, n)
    rn, rmatch, _ = orbref.sbp_keyframe(s.kp, s.desc, s.bounds, s.taken, s.K, T, SCALE, (kinds != 0).astype(np.uint8), s.pos,
                                        min_d, s.mp_desc, s.angle, 10.0, 64, True)
    nm, owner = projection_case(tmp_path, orbref, "keyframe", s, T, kinds, [0, 0, 64, 1], list(s.K) + [0.9, 10.0], None, min_d,
                                None, kp_kf)
    assert nm == rn and np.array_equal(owner, np.where(s.taken != 0, -2, rmatch)), "keyframe overload"
    total += rn
    if n >= 1000:
        assert total > 800


@pytest.mark.parametrize("kf_frame", [False, True])
def test_search_by_bow_through_the_cpp_bodies(orb, orbref, synth, tmp_path, kf_frame):
    k, L, levelsup = 10, 4, 2
    parent, desc, weight = synth.synth_vocabulary(k, L, seed=17)
    ref = orbref.RefVocabulary(parent, desc, weight, L)
    rng = np.random.default_rng(29)
    total = 0
    for n in (1500, 0, 3):
        (kp1, de1), = _frames(orbref, rng, desc, [n])
        kp2, de2 = _second_view(rng, kp1, de1)
        _, fv1 = ref.transform(de1, levelsup)
        _, fv2 = ref.transform(de2, levelsup)
        v1 = rng.choice(np.array([0, 1, 2], np.uint8), len(kp1), p=[0.1, 0.8, 0.1])
        v2 = rng.choice(np.array([0, 1, 2], np.uint8), len(kp2), p=[0.05, 0.9, 0.05])
        ratio = np.float32(0.7 if kf_frame else 0.8)
        z = np.zeros(1, np.int32)
        side = lambda kp, de, v, fv: [kp, de, v, fv[0], fv[1], fv[2] if len(fv[2]) else z[:0]]  # noqa: E731
        out = run(tmp_path, "bow_kf_f" if kf_frame else "bow_kf_kf",
                  [np.array([ratio], np.float32), np.array([1], np.int32)] + side(kp1, de1, v1, fv1) + side(kp2, de2, v2, fv2))
        nm, idx = int(np.frombuffer(out[0], np.int32)[0]), np.frombuffer(out[1], np.int32)
        if kf_frame:
            valid = (v1 == 1).astype(np.uint8)
            valid[len(kp1) - len(kp1) // 16:] = 0  # the test program cuts the keyframe's map-point vector short
            rn, rF = orbref.search_by_bow_kf_frame(kp1, de1, valid, fv1, kp2, de2, fv2, ratio, True)
            assert nm == rn and np.array_equal(idx, rF)
        else:
            rn, r12 = orbref.search_by_bow_kf_kf(kp1, de1, (v1 == 1).astype(np.uint8), fv1, kp2, de2, (v2 == 1).astype(np.uint8),
                                                 fv2, ratio, True)
            assert nm == rn and np.array_equal(idx, r12)
        total += rn
    assert total > 200


def test_distinctive_undistort_computebow_through_the_cpp_bodies(orb, orbref, synth, tmp_path):
    rng = np.random.default_rng(31)
    # MapPoint::ComputeDistinctiveDescriptors
    for nobs in (1, 2, 7, 40, 128):
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        d = np.tile(base, (nobs, 1))
        for i in range(nobs):
            for _ in range(rng.integers(0, 30)):
                d[i, rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
        out = run(tmp_path, "distinct", [d])
        assert int(np.frombuffer(out[0], np.int32)[0]) == orbref.distinctive_descriptor(d)[0]
    # Frame::UndistortKeyPoints, distorted and undistorted camera
    kp = np.zeros(700, orbref.KP_DTYPE)
    kp["x"], kp["y"] = rng.uniform(0, 640, 700).astype(np.float32), rng.uniform(0, 480, 700).astype(np.float32)
    kp["angle"], kp["octave"], kp["size"] = rng.uniform(0, 360, 700), rng.integers(0, 8, 700), 31.0
    for cam in ([517.3, 516.5, 318.6, 255.3, 0.2624, -0.9531, -0.0054, 0.0026], [500.0, 500.0, 320.0, 240.0, 0, 0, 0, 0]):
        out = run(tmp_path, "undistort", [kp, np.array(cam, np.float32)])
        got = np.frombuffer(out[0], orbref.KP_DTYPE)
        ref = orbref.undistort_keypoints(kp, cam[:4], cam[4:])
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # Frame::ComputeBoW: BowVector and FeatureVector as the containers hold them
    k, L = 9, 5
    parent, desc, weight = synth.synth_vocabulary(k, L, seed=5)
    ref = orbref.RefVocabulary(parent, desc, weight, L)
    f = _features(rng, desc, 1800)
    out = run(tmp_path, "computebow", [np.array([L], np.int32), parent.astype(np.int32), desc, weight.astype(np.float64), f])
    (rw, rv), (rn, rs, rf) = ref.transform(f, 4)
    assert np.array_equal(np.frombuffer(out[0], np.int32), rw)
    assert np.array_equal(np.frombuffer(out[1], np.float64).view(np.uint64), rv.view(np.uint64))
    assert np.array_equal(np.frombuffer(out[2], np.int32), rn)
    assert np.array_equal(np.frombuffer(out[3], np.int32), rs)
    assert np.array_equal(np.frombuffer(out[4], np.int32), rf)
    assert len(rw) > 300


def simulate_fuse(n_feat, npts, taken, kinds, host, best_idx, best_dist, sim3):
    """What the loop of ORBmatcher::Fuse leaves behind (src/ORBmatcher.cc:1044-1142, :1171-1272; MapPoint::Replace,
    src/MapPoint.cc:130-164) given bestIdx / bestDist per map point.  Map point ids as the test program numbers them:
    candidate i = i, the point a taken feature j holds on entry = npts + j."""
    slot = np.full(n_feat, -1, np.int64)
    bad = np.zeros(npts + n_feat, bool)
    obs = np.full(npts + n_feat, -1, np.int64)
    for j in np.nonzero(taken)[0]:
        slot[j] = npts + j
        obs[npts + j] = j
    bad[:npts] = kinds == 2
    for i in np.nonzero(kinds == 3)[0]:
        if host[i] >= 0:
            obs[slot[host[i]]] = -1
            slot[host[i]] = i
            obs[i] = host[i]
    found_on_entry = set(int(x) for x in slot[slot >= 0])

    def replace(this, other):  # this->Replace(other)
        if this == other:
            return
        at = obs[this]
        obs[this] = -1
        bad[this] = True
        if at >= 0:
            if obs[other] < 0:
                slot[at] = other
                obs[other] = at
            else:
                slot[at] = -1

    fused = 0
    for i in range(npts):
        if kinds[i] == 1 or bad[i]:
            continue
        if (i in found_on_entry) if sim3 else obs[i] >= 0:
            continue
        if best_dist[i] <= 20:
            j = best_idx[i]
            held = slot[j]
            if held >= 0:
                if not bad[held]:
                    if sim3:
                        replace(held, i)
                    else:
                        replace(i, held)
            else:
                if obs[i] < 0:
                    obs[i] = j
                slot[j] = i
            fused += 1
    return fused, slot, bad, obs


@pytest.mark.parametrize("sim3", [False, True])
@pytest.mark.parametrize("n", [1300, 0, 1])
def test_fuse_through_the_cpp_bodies(orb, orbref, tmp_path, n, sim3):
    """Both forms of ORBmatcher::Fuse (src/ORBmatcher.cc:1027-1145, :1147-1276): pose arithmetic on cv::Mat, the
    device search, and the Replace / AddObservation / AddMapPoint sequence on stand-in map points."""
    from test_gpu_fuse import map_point_geometry
    rng = np.random.default_rng(600 + n + sim3)
    scale = 1.3 if sim3 else 1.0
    T = pose(rng, scale=scale)
    s = Scene(orbref, rng, n, 640, 480, T, ties=True)
    normal, min_d, max_d = map_point_geometry(rng, s, T, scale)
    nf = len(s.kp)
    kinds = kinds_for(rng, n, int(s.taken.sum()), with_null=not sim3)
    host = np.full(n, -1, np.int32)
    free = list(np.nonzero(s.taken)[0])
    for i in np.nonzero(kinds == 3)[0]:
        host[i] = free.pop()
    skip = (kinds != 0).astype(np.uint8)
    rn, bi, bd, _ = orbref.fuse(s.kp, s.desc, s.bounds, s.K, SCALE, T, skip, s.pos, normal, min_d, max_d, s.mp_desc, 3.0, sim3=sim3)
    blobs = [np.zeros(4, np.int32), np.array(list(s.K) + [0.0, 3.0], np.float32), np.array(s.bounds, np.int32), s.kp, s.desc, s.taken,
             np.asarray(T, np.float32), SCALE, kinds, s.pos, normal, min_d, max_d, s.mp_desc, host]
    out = run(tmp_path, "fuse_sim3" if sim3 else "fuse_pose", blobs)
    fused = int(np.frombuffer(out[0], np.int32)[0])
    slot, isbad, obs = (np.frombuffer(out[k], np.int32) for k in (1, 2, 3))
    e_fused, e_slot, e_bad, e_obs = simulate_fuse(nf, n, s.taken, kinds, host, bi, bd, sim3)
    assert fused == e_fused == rn
    assert np.array_equal(slot, e_slot) and np.array_equal(isbad != 0, e_bad) and np.array_equal(obs, e_obs)
    if n >= 1000:
        replaced = int(e_bad.sum() - (kinds == 2).sum())
        assert fused > 250 and replaced > 20 and (e_slot[s.taken == 0] >= 0).sum() > 100


@pytest.mark.parametrize("n", [1600, 0, 1])
def test_search_by_sim3_through_the_cpp_body(orb, orbref, tmp_path, n):
    """ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1278-1516): the Sim3 algebra on cv::Mat, both device searches and the
    agreement check."""
    from test_gpu_fuse import two_views
    rng = np.random.default_rng(700 + n)
    w, h, s12, th = 752, 480, np.float32(0.93), 7.5
    T1, T2 = pose(rng), pose(rng)
    K, (v1, v2), T2s, R12, t12, mp_desc, level = two_views(orbref, rng, n, w, h, T1, T2, float(s12))
    bounds = [0, 0, w, h]
    side, kinds = [], []
    for v, other, with3 in ((v1, v2, True), (v2, v1, False)):
        k = len(v["kp"])
        has = v["owner"] >= 0
        o = np.where(has, v["owner"], 0)
        kd = rng.choice(np.array([0, 2, 3 if with3 else 0], np.uint8), k, p=[0.88, 0.06, 0.06])
        kd[~has] = 1
        pos = np.where(has[:, None], v["pos"][o] if n else np.zeros((k, 3), np.float32), 0).astype(np.float32)
        d_other = other["dist"][o] if n else np.zeros(k)
        min_d = (d_other / SCALE[level[o] if n else np.zeros(k, int)] * rng.uniform(0.85, 1.18, k)).astype(np.float32)
        max_d = (min_d * 1.2 ** 7.5).astype(np.float32)
        d = np.where(has[:, None], mp_desc[o] if n else np.zeros((k, 32), np.uint8), 0).astype(np.uint8)
        side.append([None, pos, min_d, max_d, d])
        kinds.append(kd)
    n1, n2 = len(v1["kp"]), len(v2["kp"])
    partner = np.full(n1, -1, np.int32)
    matched2 = np.zeros(n2, bool)
    for i in np.nonzero(kinds[0] == 3)[0]:
        if n2 and rng.random() < 0.8:
            partner[i] = rng.integers(0, n2)
            matched2[partner[i]] = True
    side[0][0] = (kinds[0] != 0).astype(np.uint8)
    side[1][0] = ((kinds[1] != 0) | matched2).astype(np.uint8)
    nfound, m12, _, _, _, _ = orbref.search_by_sim3(v1["kp"], v1["desc"], v2["kp"], v2["desc"], bounds, K, SCALE, T1, T2s, s12, R12,
                                                    t12, th, *side[0], *side[1])
    blobs = [np.array(list(K) + [s12, th], np.float32), np.array(bounds, np.int32), SCALE, np.asarray(T1, np.float32),
             np.asarray(T2s, np.float32), R12, t12]
    for v, sd, kd, extra in ((v1, side[0], kinds[0], partner), (v2, side[1], kinds[1], np.zeros(0, np.int32))):
        blobs += [v["kp"], v["desc"], kd, sd[1], sd[2], sd[3], sd[4], extra]
    out = run(tmp_path, "sim3search", blobs)
    got_n, got = int(np.frombuffer(out[0], np.int32)[0]), np.frombuffer(out[1], np.int32)
    assert got_n == nfound
    assert np.array_equal(got, np.where(kinds[0] == 3, -3, m12))
    if n >= 1000:
        assert nfound > 200


@pytest.mark.parametrize("n", [1500, 0, 1])
def test_search_for_triangulation_through_the_cpp_body(orb, orbref, synth, tmp_path, n):
    """ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:863-1025): FeatureVectors and map-point vectors of two
    stand-in keyframes, the fundamental matrix as a cv::Mat, and the three output vectors."""
    from test_gpu_triangulation import scene_pair
    k, L, levelsup = 10, 4, 2
    parent, desc, weight = synth.synth_vocabulary(k, L, seed=61)
    ref = orbref.RefVocabulary(parent, desc, weight, L)
    rng = np.random.default_rng(800 + n)
    v1, v2, F = scene_pair(orbref, rng, n, desc[rng.integers(1, len(desc), 50)])
    sigma2 = (SCALE * SCALE).astype(np.float32)
    n1, n2 = len(v1["kp"]), len(v2["kp"])
    _, fv1 = ref.transform(v1["desc"], levelsup)
    _, fv2 = ref.transform(v2["desc"], levelsup)
    free1, free2 = (rng.random(n1) < 0.85).astype(np.uint8), (rng.random(n2) < 0.85).astype(np.uint8)
    rn, r12 = orbref.search_for_triangulation(v1["kp"], v1["desc"], free1, fv1, v2["kp"], v2["desc"], free2, fv2, F, sigma2, True)
    z = np.zeros(0, np.int32)
    side = lambda v, fr, fv: [v["kp"], v["desc"], fr, fv[0], fv[1], fv[2] if len(fv[2]) else z]  # noqa: E731
    out = run(tmp_path, "triangulate", [F.reshape(9), sigma2, np.array([1], np.int32)] + side(v1, free1, fv1) + side(v2, free2, fv2))
    nm = int(np.frombuffer(out[0], np.int32)[0])
    pairs = np.frombuffer(out[1], np.int32).reshape(-1, 2)
    k1, k2 = np.frombuffer(out[2], orbref.KP_DTYPE), np.frombuffer(out[3], orbref.KP_DTYPE)
    i1 = np.nonzero(r12 >= 0)[0]
    assert nm == rn == len(pairs)
    assert np.array_equal(pairs[:, 0], i1) and np.array_equal(pairs[:, 1], r12[i1])
    assert np.array_equal(k1.view(np.uint8), v1["kp"][i1].view(np.uint8)) and np.array_equal(k2.view(np.uint8), v2["kp"][r12[i1]].view(np.uint8))
    if n >= 1000:
        assert rn > 100
