#!/usr/bin/env python3
"""Median latency of one 640x480 frame per call through orbhip_extract (host image in, host results out), with the
captured-graph replay (default) or plain launches (ORBHIP_GRAPH=0).  Development tool; bench.py reports the same leg."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
orb = importlib.import_module("orb-slam2_amd")
w, h, nf = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (640, 480, 1000)
ext = orb.ORBextractor(nf, 1.2, 8, 1, 20, device=0, max_width=w, max_height=h, max_batch=1)
imgs = [orb.synth.synth_image(w, h, i) for i in range(4)]
for i in range(20):
    ext(imgs[i % 4])
ts = []
for i in range(300):
    t0 = time.perf_counter()
    kp, desc = ext(imgs[i % 4])
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
print("ORBHIP_GRAPH=%s %dx%d/%d: median %.1f us, p10 %.1f, p90 %.1f, %d keypoints" %
      (os.environ.get("ORBHIP_GRAPH", "1"), w, h, nf, np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90), len(kp)))
