"""The oracle against the committed frame-level golden fixture (tests/golden/frame_golden.npz, made by
tests/golden/make_frame_golden.py): a change of the oracle, the compiler or libm that moves any output of
the frame-level path shows up here."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load(orbref):
    g = np.load(os.path.join(HERE, "golden", "frame_golden.npz"))
    kp = lambda k: np.ascontiguousarray(g[k]).view(orbref.KP_DTYPE).reshape(-1)  # noqa: E731
    return g, kp


def test_frame_golden(orbref, synth):
    g, kp = load(orbref)
    img, bounds = g["image"], [int(v) for v in g["bounds"]]
    assert np.array_equal(img, synth.synth_image(320, 240, 77)), "input generator drifted"
    kp1, d1 = orbref.RefExtractor(400, 1.2, 5, 1, 20)(img)
    assert np.array_equal(kp1.view(np.uint32).reshape(-1, 7), g["kp1"]) and np.array_equal(d1, g["desc1"])
    hk, hd = orbref.RefExtractor(400, 1.2, 5, 0, 20)(img)
    assert np.array_equal(hk.view(np.uint32).reshape(-1, 7), g["harris_kp"]) and np.array_equal(hd, g["harris_desc"])
    kp2, d2 = kp("kp2"), g["desc2"]
    cs, ci = orbref.frame_grid(kp2, bounds)
    assert np.array_equal(cs, g["grid2_start"]) and np.array_equal(ci, g["grid2_items"])
    prev = np.stack([kp1["x"], kp1["y"]], 1).astype(np.float32)
    n, m12, pv = orbref.search_for_initialization(kp1, d1, kp2, d2, bounds, prev, 30, np.float32(0.9), True)
    assert n == g["init_n"] and np.array_equal(m12, g["init_m12"]) and np.array_equal(pv, g["init_prev"])
    n, m21 = orbref.window_search(kp1, d1, g["valid1"], kp2, d2, bounds, 20, 0, 2**31 - 1, np.float32(0.8), True)
    assert n == g["window_n"] and np.array_equal(m21, g["window_m21"])
    pos = np.stack([kp1["x"], kp1["y"], np.ones(len(kp1), np.float32)], 1).astype(np.float32)
    n, mo, probes = orbref.sbp_last_frame(kp2, d2, bounds, g["taken2"], (1.0, 1.0, 0.0, 0.0), np.eye(4, dtype=np.float32),
                                          g["scale"], kp1, d1, g["skip1"], pos, 7.0, True)
    assert n == g["proj_n"] and np.array_equal(mo, g["proj_owner"])
    assert np.array_equal(probes.view(np.uint8).reshape(-1, 32), g["proj_probes"])
    voc = orbref.RefVocabulary(g["voc_parent"], g["voc_desc"], g["voc_weight"], 3)
    word, node, w = voc.transform_features(d1, 2)
    assert np.array_equal(word, g["bow_word1"]) and np.array_equal(node, g["bow_node1"]) and np.array_equal(w, g["bow_weight1"])
    cam = [float(v) for v in g["camera"]]
    un = orbref.undistort_keypoints(kp1, cam[:4], cam[4:])
    assert np.array_equal(un.view(np.uint32).reshape(-1, 7), g["undist_kp1"])
    assert orbref.image_bounds(cam[:4], cam[4:], 320, 240) == [int(v) for v in g["undist_bounds"]]
