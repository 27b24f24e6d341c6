#!/usr/bin/env python3
"""Adversarial images for the extractor (GPU box): pure noise, checkerboards, stripes, saturated blocks —
corner densities far above natural images — compared with the oracle."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orbref  # noqa: E402

orb = importlib.import_module("orb-slam2_amd")
rng = np.random.default_rng(0)
w, h = 640, 480
yy, xx = np.mgrid[0:h, 0:w]
images = {
    "noise": rng.integers(0, 256, (h, w), dtype=np.uint8),
    "noise_binary": (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8),
    "checker1": (((xx + yy) & 1) * 255).astype(np.uint8),
    "checker2": ((((xx >> 1) + (yy >> 1)) & 1) * 255).astype(np.uint8),
    "checker3": ((((xx // 3) + (yy // 3)) & 1) * 255).astype(np.uint8),
    "checker5": ((((xx // 5) + (yy // 5)) & 1) * 200 + 20).astype(np.uint8),
    "stripes_v2": (((xx >> 1) & 1) * 255).astype(np.uint8),
    "dots": np.where((xx % 4 == 0) & (yy % 4 == 0), 255, 0).astype(np.uint8),
    "dots7": np.where((xx % 7 == 3) & (yy % 7 == 3), 255, 0).astype(np.uint8),
    "ramp": ((xx * 3 + yy * 5) % 256).astype(np.uint8),
    "white": np.full((h, w), 255, np.uint8),
}
bad = 0
for name, img in images.items():
    for nf, th, st in ((1000, 20, 1), (3000, 5, 1), (1000, 20, 0)):
        tag = "%s nf=%d th=%d score=%d" % (name, nf, th, st)
        try:
            dev = orb.ORBextractor(nf, 1.2, 8, st, th, device=0, max_width=w, max_height=h)
            ref = orbref.RefExtractor(nf, 1.2, 8, st, th)
            rkp, rdesc = ref(img)
            dkp, ddesc = dev(img)
        except Exception as e:
            bad += 1
            print("ERR   %s: %s" % (tag, e), flush=True)
            continue
        ok = len(rkp) == len(dkp) and np.array_equal(rkp.view(np.uint32), dkp.view(np.uint32)) and np.array_equal(rdesc, ddesc)
        print("%s %s n=%d" % ("ok   " if ok else "FAIL ", tag, len(rkp)), flush=True)
        bad += 0 if ok else 1
print("%d bad" % bad)
sys.exit(1 if bad else 0)
