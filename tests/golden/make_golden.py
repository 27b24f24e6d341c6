#!/usr/bin/env python3
"""Generates the golden fixtures of tests/golden/ from the CPU oracle.

The reference ships no test vectors for this path and cannot be built here, so these fixtures pin
the ORACLE's behaviour (parity unpinned with respect to the reference itself): a change in the
oracle, the compiler or libstdc++/glibc that alters any keypoint, descriptor or match shows up as
a fixture mismatch.  Inputs are regenerated from seeds (orb-slam2_amd/synth.py); their SHA-256 is
stored beside the expected outputs.

  python tests/golden/make_golden.py
"""
import hashlib
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

CASES = [  # name, width, height, nfeatures, nlevels, fastTh, frame_idx
    ("f160_0", 160, 120, 200, 4, 20, 0),
    ("f160_1", 160, 120, 200, 4, 20, 1),
    ("f160_2", 160, 120, 150, 3, 20, 2),
    ("f640_0", 640, 480, 1000, 8, 20, 0),
]


def main():
    import orbref
    synth = importlib.import_module("orb-slam2_amd.synth")
    out = {}
    for name, w, h, nf, nl, th, fi in CASES:
        img = synth.synth_image(w, h, fi)
        kp, desc = orbref.RefExtractor(nf, 1.2, nl, 1, th)(img)
        out[name + "_params"] = np.array([w, h, nf, nl, th, fi], np.int32)
        out[name + "_img_sha256"] = np.frombuffer(hashlib.sha256(img.tobytes()).digest(), np.uint8)
        out[name + "_kp"] = kp.view(np.uint32).reshape(-1, 7)
        out[name + "_desc"] = desc
        print(name, len(kp), "keypoints")
    q, t = synth.synth_descriptors(300, 500, seed=42)
    idx, d1, d2 = orbref.knn2(q, t)
    out["knn_seed42_sha256"] = np.frombuffer(hashlib.sha256(q.tobytes() + t.tobytes()).digest(), np.uint8)
    out["knn_seed42"] = np.stack([idx, d1, d2])
    np.savez_compressed(os.path.join(HERE, "orb_golden.npz"), **out)


if __name__ == "__main__":
    main()
