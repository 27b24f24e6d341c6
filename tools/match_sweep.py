#!/usr/bin/env python3
"""Randomised parity sweep of the windowed matchers against the oracle (GPU box): frame sizes, keypoint
counts, window sizes, level ranges, ratios, tie-heavy and crowded frames.  usage: match_sweep.py seed cases"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orbref  # noqa: E402
import test_gpu_window as T  # noqa: E402

orb = importlib.import_module("orb-slam2_amd")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
for case in range(n_cases):
    w, h = int(rng.integers(64, 2000)), int(rng.integers(64, 1200))
    npairs = int(rng.integers(1, 5))
    sizes = [int(rng.choice([0, 1, 2, 63, 64, 65, int(rng.integers(1, 6000))])) for _ in range(npairs)]
    levels = int(rng.integers(1, 9))
    ties, cluster, ori = bool(rng.random() < 0.5), bool(rng.random() < 0.5), bool(rng.random() < 0.7)
    window = int(rng.choice([1, 5, 10, 30, 100, 200, 500]))
    ratio = float(rng.choice([0.5, 0.6, 0.75, 0.9, 0.95, 1.0]))
    lo = int(rng.integers(-1, levels))
    hi = T.INT_MAX if rng.random() < 0.5 else lo + int(rng.integers(0, 3))
    seed = int(rng.integers(0, 100000))
    bounds = [0, 0, w, h] if rng.random() < 0.7 else [-int(rng.integers(0, 20)), -int(rng.integers(0, 20)), w + 9, h + 5]
    tag = "%dx%d sizes=%s levels=%d ties=%d cluster=%d ori=%d window=%d ratio=%.2f lo=%d hi=%d seed=%d bounds=%s" % (
        w, h, sizes, levels, ties, cluster, ori, window, ratio, lo, hi, seed, bounds)
    fr, r2 = T.frames_for(orbref, seed, sizes, w, h, levels=levels, ties=ties, cluster=cluster)
    pairs, valids = [], []
    for kp, de in fr:
        pairs.append((kp, de) + T.moved(orbref, r2, kp, de, w, h, rot=float(r2.uniform(0, 360)), shift=float(r2.uniform(0, 12))))
        valids.append((r2.random(len(kp)) < 0.7).astype(np.uint8))
    for name, fn in (("init", lambda: T._run_search_init(orb, orbref, None, pairs, bounds, window, ratio, ori, cap=T._pad_equal(pairs))),
                     ("window", lambda: T._run_window(orb, orbref, pairs, valids, bounds, window, lo, hi, ratio, ori))):
        try:
            nm = fn()
            print("ok    %-6s %s matched=%d" % (name, tag, nm), flush=True)
        except AssertionError as e:
            bad += 1
            print("FAIL  %-6s %s: %s" % (name, tag, e), flush=True)
        except orb.OrbhipError as e:
            print("ERR   %-6s %s: %s" % (name, tag, e), flush=True)
print("%d failing of %d" % (bad, 2 * n_cases))
sys.exit(1 if bad else 0)
