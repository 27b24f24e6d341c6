"""The device path against the committed golden fixtures alone (tests/golden/*.npz): no oracle call in this
file.  orb_golden.npz: extraction of four seeded frames and one kNN problem; frame_golden.npz: Harris-score
extraction, feature grid, windowed matchers, projection matchers, BoW and undistortion on stored inputs.
Bar: bit-exact."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_extraction_and_knn_golden(orb, synth):
    g = np.load(os.path.join(HERE, "golden", "orb_golden.npz"))
    for name in sorted(k[:-7] for k in g.files if k.endswith("_params")):
        w, h, nf, nl, th, fi = (int(v) for v in g[name + "_params"])
        img = synth.synth_image(w, h, fi)
        assert hashlib.sha256(img.tobytes()).digest() == g[name + "_img_sha256"].tobytes()
        kp, desc = orb.ORBextractor(nf, 1.2, nl, 1, th, device=0, max_width=w, max_height=h)(img)
        assert np.array_equal(kp.view(np.uint32).reshape(-1, 7), g[name + "_kp"]), name
        assert np.array_equal(desc, g[name + "_desc"]), name
    q, t = synth.synth_descriptors(300, 500, seed=42)
    idx, d1, d2 = orb.ORBmatcher().knn2(q, t)
    assert np.array_equal(np.stack([idx, d1, d2]), g["knn_seed42"])


def test_frame_level_golden(orb):
    import torch
    g = np.load(os.path.join(HERE, "golden", "frame_golden.npz"))
    kp = lambda k: np.ascontiguousarray(g[k]).view(orb.KP_DTYPE).reshape(-1)  # noqa: E731
    img, bounds = g["image"], [int(v) for v in g["bounds"]]
    k1, d1 = orb.ORBextractor(400, 1.2, 5, 1, 20, device=0, max_width=320, max_height=240)(img)
    assert np.array_equal(k1.view(np.uint32).reshape(-1, 7), g["kp1"]) and np.array_equal(d1, g["desc1"])
    hk, hd = orb.ORBextractor(400, 1.2, 5, 0, 20, device=0, max_width=320, max_height=240)(img)
    assert np.array_equal(hk.view(np.uint32).reshape(-1, 7), g["harris_kp"]) and np.array_equal(hd, g["harris_desc"])
    kp1, kp2, d2 = kp("kp1"), kp("kp2"), g["desc2"]
    n1, n2 = len(kp1), len(kp2)
    m = orb.ORBmatcher(0.9, True)
    F1 = orb.FrameBatch.from_host(m, [kp1], [d1], bounds, cap=max(n1, n2))
    F2 = orb.FrameBatch.from_host(m, [kp2], [d2], bounds, cap=max(n1, n2))
    cs, ci = F2.grid(0)
    assert np.array_equal(cs, g["grid2_start"]) and np.array_equal(ci, g["grid2_items"])
    prev = np.zeros((1, F1.cap, 2), np.float32)
    prev[0, :n1, 0], prev[0, :n1, 1] = kp1["x"], kp1["y"]
    d_prev = torch.from_numpy(prev).cuda()
    nm, m12 = m.SearchForInitialization(F1, F2, d_prev, 30)
    assert int(nm[0]) == g["init_n"] and np.array_equal(m12[0, :n1].cpu().numpy(), g["init_m12"])
    assert np.array_equal(d_prev[0, :n1].cpu().numpy(), g["init_prev"])
    m8 = orb.ORBmatcher(0.8, True)
    valid = np.zeros((1, F1.cap), np.uint8)
    valid[0, :n1] = g["valid1"]
    nm, m21 = m8.WindowSearch(F1, F2, 20, torch.from_numpy(valid).cuda(), 0)
    assert int(nm[0]) == g["window_n"] and np.array_equal(m21[0, :n2].cpu().numpy(), g["window_m21"])
    taken = np.zeros((1, F2.cap), np.uint8)
    taken[0, :n2] = g["taken2"]
    d_taken, d_np = torch.from_numpy(taken).cuda(), torch.tensor([n1], dtype=torch.int32, device="cuda")
    d_desc1 = torch.from_numpy(d1[None].copy()).cuda()
    for key, rule, matcher in (("proj", orb.ORBmatcher.PROJ_BEST, m), ("local", orb.ORBmatcher.PROJ_LOCAL_MAP, m8)):
        probes = torch.from_numpy(g[key + "_probes"][None].copy()).cuda()
        nm, owner = matcher.SearchByProjection(F2, probes, d_desc1, d_np, rule, taken=d_taken, th_dist=100,
                                               check_orientation=(key == "proj"))
        assert int(nm[0]) == g[key + "_n"], key
        assert np.array_equal(owner[0, :n2].cpu().numpy(), g[key + "_owner"]), key
    voc = orb.ORBVocabulary(g["voc_parent"], g["voc_desc"], g["voc_weight"], 3)
    word, node, w = voc.transform_features(m, d1, 2)
    assert np.array_equal(word, g["bow_word1"]) and np.array_equal(node, g["bow_node1"]) and np.array_equal(w, g["bow_weight1"])
    F1.ComputeBoW(voc, 2, node_cap=64)
    F2.ComputeBoW(voc, 2, node_cap=64)
    fn, fs, ff = F1.feature_vector(0)
    assert np.array_equal(fn, g["fv1_nodes"]) and np.array_equal(fs, g["fv1_start"]) and np.array_equal(ff, g["fv1_feat"])
    v2 = np.zeros((1, F2.cap), np.uint8)
    v2[0, :n2] = 1
    nm, m12b, _ = m8.SearchByBoW(F1, F2, torch.from_numpy(valid).cuda(), torch.from_numpy(v2).cuda(), kf_frame=False)
    assert int(nm[0]) == g["bow_n"] and np.array_equal(m12b[0, :n1].cpu().numpy(), g["bow_m12"])
    cam = [float(v) for v in g["camera"]]
    un = orb.UndistortKeyPoints(m, F1.kp, F1.n, cam[:4], cam[4:]).cpu().numpy()
    got = np.ascontiguousarray(un.reshape(-1)[:n1 * 7]).view(np.uint32).reshape(n1, 7)
    assert np.array_equal(got, g["undist_kp1"])
    assert orb.ComputeImageBounds(cam[:4], cam[4:], 320, 240) == [int(v) for v in g["undist_bounds"]]
