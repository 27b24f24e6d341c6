#!/usr/bin/env python3
"""Step time of the 256 x 1280x720 workload against the extractor's chunk size (frames that go through all
stages together): smaller chunks keep a stage's output in the last-level cache for the next stage."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
orb = importlib.import_module("orb-slam2_amd")
import torch  # noqa: E402

b, w, h, nf = 256, 1280, 720, 2000
frames = np.stack([orb.synth.synth_image(w, h, i) for i in range(32)] * 8)
d_img = torch.from_numpy(frames).cuda()
d_kp = torch.zeros((b, nf, 7), dtype=torch.float32, device="cuda")
d_desc = torch.zeros((b, nf, 32), dtype=torch.uint8, device="cuda")
d_n = torch.zeros(b, dtype=torch.int32, device="cuda")
for mb in (int(a) for a in sys.argv[1:]):
    ext = orb.ORBextractor(nf, 1.2, 8, 1, 20, device=0, max_width=w, max_height=h, max_batch=mb)
    def step():
        ext.extract_batch_device(d_img.data_ptr(), b, h, w, w, w * h, d_kp.data_ptr(), d_desc.data_ptr(), nf, d_n.data_ptr())
    for _ in range(3):
        step()
    ext.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    ext.synchronize()
    print("max_batch %4d: %.3f ms per 256 frames" % (mb, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
    ext.close()
