#!/usr/bin/env python3
"""Randomised parity sweep of the extractor against the oracle (GPU box): image sizes, feature counts,
level counts, scale factors, thresholds and both score types.  Prints the failing configurations."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orbref  # noqa: E402

orb = importlib.import_module("orb-slam2_amd")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
max_w = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
max_h = int(sys.argv[4]) if len(sys.argv) > 4 else 800
bad = 0
for case in range(n_cases):
    w, h = int(rng.integers(96, max_w)), int(rng.integers(96, max_h))
    nf = int(rng.integers(50, 3000 * max(1, max_w // 1000)))
    nl = int(rng.integers(1, 9))
    sf = float(rng.choice([1.1, 1.2, 1.2, 1.3, 1.5, 2.0]))
    th = int(rng.integers(5, 45))
    st = int(rng.integers(0, 2))
    seed = int(rng.integers(0, 10000))
    img = orb.synth.synth_image(w, h, seed)
    flat = rng.random() < 0.2
    if flat:
        img = (img // 8 * 8).astype(np.uint8)  # flatter image: threshold-7 fallback and sparse cells
    tag = "%d %d %d %d %.1f %d %d %d%s" % (w, h, nf, nl, sf, th, st, seed, " flat" if flat else "")
    try:
        dev = orb.ORBextractor(nf, sf, nl, st, th, device=0, max_width=w, max_height=h)
    except orb.OrbhipError as e:
        print("SKIP  %s: %s" % (tag, e))
        continue
    ref = orbref.RefExtractor(nf, sf, nl, st, th)
    try:
        rkp, rdesc = ref(img)
        dkp, ddesc = dev(img)
    except Exception as e:  # geometry the reference cannot run either
        print("ERR   %s: %s" % (tag, e))
        continue
    ok = len(rkp) == len(dkp) and np.array_equal(rkp.view(np.uint32), dkp.view(np.uint32)) and np.array_equal(rdesc, ddesc)
    print("%s %s n=%d" % ("ok   " if ok else "FAIL ", tag, len(rkp)), flush=True)
    bad += 0 if ok else 1
    dev.close()
print("%d failing of %d" % (bad, n_cases))
sys.exit(1 if bad else 0)
