"""Parity of the device form of ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:863-1025, with
CheckDistEpipolarLine :139-156) against the oracle's literal restatement (oracle/bow_ref.cpp), through the C ABI:
two keyframes of one synthetic scene, their FeatureVectors from a synthetic vocabulary, the fundamental matrix of
the two poses.  Map-point descriptors come from a small pool, so a feature has several candidates within TH_LOW
under its node and the epipolar line, the 2 * BestDist bound and the taken flags decide.  Bar: bit-exact."""
import numpy as np
import pytest

from test_gpu_fuse import two_views
from test_gpu_projection import SCALE, pose

pytestmark = pytest.mark.gpu


def fundamental(K, R12, t12):
    """F12 with x1' F12 x2 = 0 for p_c1 = R12 p_c2 + t12 (LocalMapping::ComputeF12, src/LocalMapping.cc)."""
    Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float64)
    t = t12.astype(np.float64)
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Ki = np.linalg.inv(Km)
    return (Ki.T @ tx @ R12.astype(np.float64) @ Ki).astype(np.float32)


def scene_pair(orbref, rng, n, pool):
    T1, T2 = pose(rng), pose(rng)
    T2[:3, 3] += np.array([0.6, 0.1, 0.0], np.float32)  # a baseline, so that the epipolar lines mean something
    K, (v1, v2), _, R12, t12, _, _ = two_views(orbref, rng, n, 752, 480, T1, T2, 1.0, pool=pool)
    # the two views of a point differ by a common image rotation (plus noise and a few wild angles), as real keyframes do
    base = rng.uniform(0, 360, max(n, 1))
    for v, offset in ((v1, 0.0), (v2, -25.0)):
        has = v["owner"] >= 0
        a = base[np.where(has, v["owner"], 0)] + offset + rng.normal(0, 6, len(has))
        wild = rng.random(len(has)) < 0.1
        v["kp"]["angle"] = np.where(has & ~wild, np.mod(a, 360), v["kp"]["angle"]).astype(np.float32)
    return v1, v2, fundamental(K, R12, t12)


@pytest.mark.parametrize("ori", [True, False])
def test_search_for_triangulation(orb, orbref, synth, ori):
    import torch
    k, L, levelsup = 10, 4, 2
    parent, desc, weight = synth.synth_vocabulary(k, L, seed=41)
    voc = orb.ORBVocabulary(parent, desc, weight, L)
    ref = orbref.RefVocabulary(parent, desc, weight, L)
    rng = np.random.default_rng(43 + ori)
    pool = desc[rng.integers(1, len(desc), 60)]
    pairs = [scene_pair(orbref, rng, n, pool) for n in (1800, 0, 1, 300, 2500)]
    sigma2 = (SCALE * SCALE).astype(np.float32)
    m = orb.ORBmatcher(0.6, ori)
    B = len(pairs)
    cap = max(max(len(p[0]["kp"]), len(p[1]["kp"])) for p in pairs) or 1
    F1 = orb.FrameBatch.from_host(m, [p[0]["kp"] for p in pairs], [p[0]["desc"] for p in pairs], [0, 0, 752, 480], cap=cap)
    F2 = orb.FrameBatch.from_host(m, [p[1]["kp"] for p in pairs], [p[1]["desc"] for p in pairs], [0, 0, 752, 480], cap=cap)
    F1.ComputeBoW(voc, levelsup, node_cap=256)
    F2.ComputeBoW(voc, levelsup, node_cap=256)
    free1 = np.zeros((B, cap), np.uint8)
    free2 = np.zeros((B, cap), np.uint8)
    for f, (v1, v2, _) in enumerate(pairs):
        free1[f, :len(v1["kp"])] = rng.random(len(v1["kp"])) < 0.85
        free2[f, :len(v2["kp"])] = rng.random(len(v2["kp"])) < 0.85
    F12 = np.stack([p[2] for p in pairs])
    nm, m12, m21 = m.SearchForTriangulation(F1, F2, torch.from_numpy(free1).cuda(), torch.from_numpy(free2).cuda(),
                                            torch.from_numpy(F12).cuda(), torch.from_numpy(sigma2).cuda())
    m.synchronize()
    nm, m12, m21 = nm.cpu().numpy(), m12.cpu().numpy(), m21.cpu().numpy()
    total = true_pairs = 0
    for f, (v1, v2, F) in enumerate(pairs):
        n1, n2 = len(v1["kp"]), len(v2["kp"])
        _, fv1 = ref.transform(v1["desc"], levelsup)
        _, fv2 = ref.transform(v2["desc"], levelsup)
        rn, r12 = orbref.search_for_triangulation(v1["kp"], v1["desc"], free1[f, :n1], fv1, v2["kp"], v2["desc"], free2[f, :n2], fv2,
                                                  F, sigma2, ori)
        assert np.array_equal(r12, m12[f, :n1]), "pair %d: %d entries of vMatches12 differ" % (f, int((r12 != m12[f, :n1]).sum()))
        assert rn == nm[f], "pair %d nmatches" % f
        assert (m12[f, n1:] == -1).all(), "pair %d: entries past the keyframe's features" % f
        for i1 in np.nonzero(m12[f] >= 0)[0]:
            assert m21[f, m12[f, i1]] == i1, "pair %d: matches21 is not the inverse of matches12" % f
        assert (m21[f] >= 0).sum() == (m12[f] >= 0).sum() == nm[f], "pair %d: counts" % f
        ok = r12 >= 0
        true_pairs += int((v1["owner"][ok] == v2["owner"][r12[ok]]).sum())
        total += rn
    assert total > 400 and true_pairs > 0.5 * total


def test_search_for_triangulation_host_form_and_errors(orb, orbref, synth):
    k, L, levelsup = 9, 3, 1
    parent, desc, weight = synth.synth_vocabulary(k, L, seed=47)
    ref = orbref.RefVocabulary(parent, desc, weight, L)
    rng = np.random.default_rng(53)
    v1, v2, F = scene_pair(orbref, rng, 900, desc[rng.integers(1, len(desc), 30)])
    sigma2 = (SCALE * SCALE).astype(np.float32)
    _, fv1 = ref.transform(v1["desc"], levelsup)
    _, fv2 = ref.transform(v2["desc"], levelsup)
    n1, n2 = len(v1["kp"]), len(v2["kp"])
    free1, free2 = np.ones(n1, np.uint8), (rng.random(n2) < 0.9).astype(np.uint8)
    rn, r12 = orbref.search_for_triangulation(v1["kp"], v1["desc"], free1, fv1, v2["kp"], v2["desc"], free2, fv2, F, sigma2, True)
    m = orb.ORBmatcher(0.6, True)
    L_ = orb._lib.lib()
    C = orb._lib.C
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    arr = lambda fv: [np.ascontiguousarray(x if len(x) else np.zeros(1), np.int32) for x in fv]  # noqa: E731
    a, b = arr(fv1), arr(fv2)
    out = np.full(n1, -7, np.int32)
    nm = C.c_int32(-1)
    kp1, kp2 = np.ascontiguousarray(v1["kp"]), np.ascontiguousarray(v2["kp"])
    d1, d2 = np.ascontiguousarray(v1["desc"]), np.ascontiguousarray(v2["desc"])
    Ff = np.ascontiguousarray(F.reshape(9))
    call = lambda s2, nlev: L_.orbhip_search_for_triangulation(  # noqa: E731
        m._h, p(kp1), p(d1), n1, p(free1), p(a[0]), p(a[1]), p(a[2]), len(fv1[0]), p(kp2), p(d2), n2, p(free2), p(b[0]), p(b[1]),
        p(b[2]), len(fv2[0]), p(Ff), p(s2), nlev, 1, p(out), C.byref(nm))
    orb._lib.check(call(sigma2, len(sigma2)))
    assert nm.value == rn and np.array_equal(out, r12) and rn > 100
    # an octave of keyframe 2 beyond the sigma table is refused, not clamped silently
    with pytest.raises(orb.OrbhipError):
        orb._lib.check(call(sigma2[:3].copy(), 3))
