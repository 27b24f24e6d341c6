"""Frame::ComputeImageBounds through the C ABI (host computation, no GPU) against the oracle."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def test_image_bounds_match_the_oracle():
    import orbref
    orb = importlib.import_module("orb-slam2_amd")
    rng = np.random.default_rng(9)
    cams = [((517.3, 516.5, 318.6, 255.3), (0.2624, -0.9531, -0.0054, 0.0026)),   # TUM1-like
            ((535.4, 539.2, 320.1, 247.6), (0.0, 0.0, 0.0, 0.0)),
            ((458.654, 457.296, 367.215, 248.375), (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05))]
    for _ in range(50):
        cams.append((tuple(rng.uniform(300, 900, 2)) + tuple(rng.uniform(200, 400, 2)),
                     tuple(rng.uniform(-0.3, 0.3, 2)) + tuple(rng.uniform(-0.01, 0.01, 2))))
    for K, D in cams:
        for cols, rows in ((640, 480), (752, 480), (1280, 720)):
            assert orb.ComputeImageBounds(K, D, cols, rows) == orbref.image_bounds(K, D, cols, rows), (K, D, cols, rows)
