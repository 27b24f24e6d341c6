#!/usr/bin/env python3
"""Randomised parity sweep of the projection matchers (last-frame, keyframe, local-map, frame-to-frame and Sim3
overloads) against the oracle (GPU box).  usage: projection_sweep.py seed cases"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orbref  # noqa: E402
import test_gpu_projection as T  # noqa: E402

orb = importlib.import_module("orb-slam2_amd")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
for case in range(n_cases):
    w, h = int(rng.integers(120, 1600)), int(rng.integers(100, 1000))
    n = int(rng.choice([0, 1, 2, 65, int(rng.integers(1, 5000))]))
    ties, big = bool(rng.random() < 0.5), bool(rng.random() < 0.2)
    kind = int(rng.integers(0, 5))
    ori = bool(rng.random() < 0.6)
    ratio = float(rng.choice([0.6, 0.8, 0.9, 1.0]))
    th = float(rng.choice([1.0, 3.0, 7.0, 15.0, 40.0]))
    tag = "kind=%d %dx%d n=%d ties=%d big=%d ori=%d ratio=%.1f th=%.0f" % (kind, w, h, n, ties, big, ori, ratio, th)
    try:
        T_ = T.pose(rng, scale=1.3 if kind == 4 else 1.0, big=big)
        s = T.Scene(orbref, rng, n, w, h, T_, ties=ties)
        skip = (rng.random(n) < 0.15).astype(np.uint8)
        if kind == 0:
            kp_last = np.zeros(n, orbref.KP_DTYPE)
            kp_last["octave"], kp_last["angle"] = s.level, s.angle
            r = orbref.sbp_last_frame(s.kp, s.desc, s.bounds, s.taken, s.K, T_, T.SCALE, kp_last, s.mp_desc, skip, s.pos, th, ori)
            rule, td, o = orb.ORBmatcher.PROJ_BEST, 100, ori
        elif kind == 1:
            R = T_[:3, :3].astype(np.float64)
            Ow = -R.T @ T_[:3, 3].astype(np.float64)
            dist = np.linalg.norm(s.pos.astype(np.float64) - Ow, axis=1) if n else np.zeros(0)
            min_d = (dist / T.SCALE[s.level] * rng.uniform(0.9, 1.1, n)).astype(np.float32)
            td = int(rng.choice([50, 64, 100]))
            r = orbref.sbp_keyframe(s.kp, s.desc, s.bounds, s.taken, s.K, T_, T.SCALE, skip, s.pos, min_d, s.mp_desc, s.angle, th, td, ori)
            rule, o = orb.ORBmatcher.PROJ_BEST, ori
        elif kind == 2:
            cam = s.pos.astype(np.float64) @ T_[:3, :3].T.astype(np.float64) + T_[:3, 3]
            px = (s.K[0] * cam[:, 0] / cam[:, 2] + s.K[2]).astype(np.float32) if n else np.zeros(0, np.float32)
            py = (s.K[1] * cam[:, 1] / cam[:, 2] + s.K[3]).astype(np.float32) if n else np.zeros(0, np.float32)
            cosv = rng.choice(np.array([0.9, 0.998, 0.9981, 1.0], np.float32), n)
            r = orbref.sbp_local_map(s.kp, s.desc, s.bounds, T.SCALE, s.taken, 1 - skip, (rng.random(n) < 0.05).astype(np.uint8),
                                     s.level, cosv, px, py, s.mp_desc, min(th, 3.0), ratio)
            rule, td, o = orb.ORBmatcher.PROJ_LOCAL_MAP, 100, False
        elif kind == 3:
            kp1 = np.zeros(n, orbref.KP_DTYPE)
            kp1["octave"] = s.level
            r = orbref.sbp_frames(kp1, s.mp_desc, skip, s.pos, T_, s.K, s.kp, s.desc, s.bounds, s.taken, int(th) + 5, ratio)
            rule, td, o = orb.ORBmatcher.PROJ_RATIO, 100, False
        else:
            R = T_[:3, :3].astype(np.float64) / 1.3
            Ow = -R.T @ (T_[:3, 3].astype(np.float64) / 1.3)
            PO = s.pos.astype(np.float64) - Ow
            dist = np.linalg.norm(PO, axis=1) if n else np.zeros(0)
            normal = (PO / np.maximum(dist[:, None], 1e-9) + rng.normal(0, 0.5, (n, 3))).astype(np.float32)
            min_d = (dist / 1.2 ** rng.uniform(-1, 8, n)).astype(np.float32)
            r = orbref.sbp_sim3(s.kp, s.desc, s.bounds, s.K, T.SCALE, T_, s.taken, skip, s.pos, normal, min_d, min_d * 4, s.mp_desc, int(th) + 3)
            rule, td, o = orb.ORBmatcher.PROJ_BEST, 20, False
        nm = T.run_device(orb, [s], [r + (s.mp_desc, s.taken)], rule, ratio=ratio, th_dist=td, ori=o)
        print("ok    %s matched=%d" % (tag, nm), flush=True)
    except AssertionError as e:
        bad += 1
        print("FAIL  %s: %s" % (tag, e), flush=True)
print("%d failing of %d" % (bad, n_cases))
sys.exit(1 if bad else 0)
