#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: the HIP extractor and the Hamming scan against the CPU oracle on
configurations drawn from a seeded generator, the two windowed matchers on random frame pairs, the five SearchByProjection overloads, both Fuse forms, SearchBySim3 and
SearchForTriangulation on reseeded scenes, for
a time budget.  Everything is compared bit for bit (keypoint
fields including the orientation's float bits, descriptors, kNN indices and distances).  Writes one JSON summary.
   python3 tools/parity_sweep.py [--seconds 240] [--seed 1] [--out gpurun_out/parity_sweep.json]
A geometry whose cells reach past the border must be refused by both sides (the reference would read outside
the image there); anything else the device refuses is counted separately."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def image_of(kind, rng, synth, w, h, idx):
    if kind == "scene":
        return synth.synth_image(w, h, idx)
    if kind == "scene_flat":
        return (synth.synth_image(w, h, idx) // 8 * 8).astype(np.uint8)
    if kind == "noise":
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == "smooth":
        yy, xx = np.mgrid[0:h, 0:w]
        a, b = rng.uniform(0.02, 0.2, 2)
        return np.clip(127 + 100 * np.sin(a * xx + rng.uniform(0, 6)) * np.cos(b * yy) + rng.integers(-3, 4, (h, w)), 0, 255).astype(np.uint8)
    if kind == "blocks":
        s = int(rng.integers(2, 9))
        small = rng.integers(0, 256, ((h + s - 1) // s, (w + s - 1) // s), dtype=np.uint8)
        return np.kron(small, np.ones((s, s), np.uint8))[:h, :w].copy()
    raise ValueError(kind)


def same_extraction(rkp, rdesc, dkp, ddesc):
    if len(rkp) != len(dkp):
        return "count %d vs %d" % (len(rkp), len(dkp))
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        if not np.array_equal(rkp[f], dkp[f]):
            return f
    if not np.array_equal(rkp["angle"].view(np.uint32), dkp["angle"].view(np.uint32)):
        return "angle bits"
    if not np.array_equal(rdesc, ddesc):
        return "%d descriptor bits" % int(np.unpackbits(rdesc ^ ddesc).sum())
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_sweep.json"))
    args = ap.parse_args()
    import orbref
    orb = importlib.import_module("orb-slam2_amd")
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    ex = {"configurations": 0, "frames": 0, "keypoints": 0, "refused_geometries": 0, "unsupported_by_device": 0, "mismatching_frames": 0,
          "failures": []}
    kn = {"calls": 0, "comparisons": 0, "mismatching_calls": 0, "failures": []}
    wm = {"pairs": 0, "mismatching_calls": 0, "failures": []}  # SearchForInitialization + WindowSearch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_window as wt  # frame generators and the compare-with-the-oracle helpers of the tests
    INT_MAX = 2147483647
    import test_gpu_projection as tp  # scenes, oracle calls and the device-against-oracle comparison of the tests
    pj = {"calls": 0, "mismatching_calls": 0, "failures": []}  # SearchByProjection, all five overloads
    import test_gpu_fuse as tf
    import test_gpu_triangulation as tt
    import test_gpu_probes as tb
    fm = {"calls": 0, "mismatching_calls": 0, "failures": []}  # Fuse x2, SearchBySim3, SearchForTriangulation, the probe builder
    real_default_rng = np.random.default_rng
    real_run_device = tp.run_device

    class ProjectionMismatch(Exception):
        pass

    def checked_run_device(*a, **k):
        try:
            return real_run_device(*a, **k)
        except AssertionError as e:
            raise ProjectionMismatch(str(e) or "device result differs from the oracle's")
    tp.run_device = checked_run_device
    kinds = ["scene", "scene", "scene_flat", "noise", "smooth", "blocks"]
    last_print = time.time()
    while time.time() < t_end:
        # ---- one extractor configuration, a few frames through it (single and batched) -------------
        w, h = int(rng.integers(96, 1400)), int(rng.integers(96, 800))
        nf = int(rng.integers(50, 4001))
        nl = int(rng.integers(1, 9))
        sf = float(rng.choice([1.1, 1.2, 1.2, 1.3, 1.5, 2.0]))
        th = int(rng.integers(5, 45))
        st = int(rng.integers(0, 2))
        gauss = int(rng.integers(0, 2))
        cfg = dict(w=w, h=h, nfeatures=nf, nlevels=nl, scale=sf, fast_th=th, score=st, gauss_profile=gauss)
        imgs = [image_of(str(rng.choice(kinds)), rng, orb.synth, w, h, int(rng.integers(0, 100000))) for _ in range(3)]
        dev, dev_err, ref_err, outs, refs = None, None, None, None, None
        try:
            dev = orb.ORBextractor(nf, sf, nl, st, th, device=0, max_width=w, max_height=h, max_batch=3)
            dev.set_gaussian_profile(gauss)
            outs = [dev(imgs[0])] + list(dev.extract_batch(np.stack(imgs))[1:])
        except orb.OrbhipError as e:
            dev_err = e
        finally:
            if dev is not None:
                dev.close()
        orbref.lib().orbref_set_gauss_profile(gauss)
        try:
            ref = orbref.RefExtractor(nf, sf, nl, st, th)
            refs = [ref(im) for im in imgs]
        except RuntimeError as e:
            ref_err = e
        finally:
            orbref.lib().orbref_set_gauss_profile(0)
        if dev_err is not None or ref_err is not None:
            # a geometry whose cells reach past the border (the reference would read outside the image there) must
            # be refused by both; other device refusals (sizes it does not support) are only counted
            geom = dev_err is not None and dev_err.code == -5
            if (ref_err is not None) != geom:
                ex["mismatching_frames"] += 1
                ex["failures"].append(dict(cfg, what="refusal: device %r, oracle %r" % (dev_err, ref_err)))
            ex["refused_geometries" if geom else "unsupported_by_device"] += 1
            continue
        ex["configurations"] += 1
        for (rkp, rdesc), (dkp, ddesc) in zip(refs, outs):
            ex["frames"] += 1
            ex["keypoints"] += len(rkp)
            bad = same_extraction(rkp, rdesc, dkp, ddesc)
            if bad:
                ex["mismatching_frames"] += 1
                ex["failures"].append(dict(cfg, what=bad))
        # ---- a few Hamming scans of random shape through both kernels -------------------------------
        for scan in ("matrix", "popcount"):
            nq, nt = int(rng.integers(1, 700)), int(rng.choice([rng.integers(1, 100), rng.integers(100, 6000), rng.integers(6000, 60000)]))
            pairs = int(rng.integers(1, 4)) if nt < 6000 else 1
            q = rng.integers(0, 256, (pairs, nq, 32), dtype=np.uint8)
            t = rng.integers(0, 256, (pairs, nt, 32), dtype=np.uint8)
            if rng.random() < 0.5:  # near-duplicates: small distances and ties
                k = int(rng.integers(1, min(nq, nt) + 1))
                t[:, rng.integers(0, nt, k)] = q[:, rng.integers(0, nq, k)]
                if nt > 3:
                    t[:, nt - 1] = t[:, 0]
            m = orb.ORBmatcher(0.6, True)
            m.set_scan_kernel(scan)
            idx, d1, d2 = m.knn2(q, t)
            ok = True
            for p in range(pairs):
                ridx, rd1, rd2 = orbref.knn2(q[p], t[p])
                ok = ok and np.array_equal(idx[p], ridx) and np.array_equal(d1[p], rd1) and np.array_equal(d2[p], rd2)
            kn["calls"] += 1
            kn["comparisons"] += pairs * nq * nt
            if not ok:
                kn["mismatching_calls"] += 1
                kn["failures"].append(dict(scan=scan, nq=nq, nt=nt, pairs=pairs))
            m.close()
        # ---- the two windowed matchers on a few random frame pairs ------------------------------------------
        try:
            ww, wh = int(rng.integers(200, 1400)), int(rng.integers(160, 800))
            levels = int(rng.integers(1, 6))
            ties, cluster, ori = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            sizes = [int(rng.integers(0, 60)), int(rng.integers(60, 900)), int(rng.integers(900, 2600))]
            fr, frng = wt.frames_for(orbref, int(rng.integers(0, 1 << 30)), sizes, ww, wh, levels=levels, ties=ties, cluster=cluster)
            pairs, valids = [], []
            for kp, de in fr:
                pairs.append((kp, de) + wt.moved(orbref, frng, kp, de, ww, wh, noise_bits=int(rng.integers(0, 8)),
                                                 shift=float(rng.uniform(1, 12)), rot=float(rng.uniform(0, 360))))
                valids.append((frng.random(len(kp)) < 0.7).astype(np.uint8))
            window = int(rng.choice([10, 30, 50, 100, 200]))
            ratio = float(rng.choice([0.6, 0.75, 0.9]))
            bounds = [0, 0, ww, wh]
            wm["pairs"] += 2 * len(pairs)
            wt._run_search_init(orb, orbref, None, pairs, bounds, window, ratio, ori, cap=wt._pad_equal(pairs))
            lo = int(rng.integers(-1, levels))
            wt._run_window(orb, orbref, pairs, valids, bounds, window, lo, INT_MAX if rng.random() < 0.7 else lo + 1, ratio, ori)
        except AssertionError as e:
            wm["mismatching_calls"] += 1
            wm["failures"].append(dict(w=ww, h=wh, levels=levels, ties=ties, cluster=cluster, ori=ori, sizes=sizes, what=str(e)[:200]))
        # ---- the five SearchByProjection overloads: the scenes of tests/test_gpu_projection.py under fresh seeds ----
        it_seed = int(rng.integers(0, 1 << 30))
        np.random.default_rng = lambda seed=None, _s=it_seed: real_default_rng(None if seed is None else [int(seed), _s])
        try:
            for fn, kw in ((tp.test_local_map_overload, dict(th=float(rng.choice([1.0, 3.0])), ties=bool(rng.integers(0, 2)))),
                           (tp.test_sim3_overload, dict(ties=bool(rng.integers(0, 2)))),
                           (tp.test_frame_to_frame_overload, dict(window=int(rng.choice([15, 60])), ratio=float(rng.choice([0.8, 0.95])))),
                           (tp.test_last_frame_overload, dict(th=float(rng.choice([7.0, 15.0])), ori=bool(rng.integers(0, 2)))),
                           (tp.test_keyframe_overload, dict(orb_dist=int(rng.choice([64, 100])), ori=bool(rng.integers(0, 2))))):
                pj["calls"] += 1
                try:
                    fn(orb, orbref, **kw)
                except ProjectionMismatch as e:
                    pj["mismatching_calls"] += 1
                    pj["failures"].append(dict(test=fn.__name__, seed=it_seed, what=str(e)[:200], **kw))
                except AssertionError:
                    pass  # the tests' own "enough matches" checks are tuned to their fixed seeds, not parity statements
            # ---- Fuse (both forms), SearchBySim3, SearchForTriangulation: the scenes of their tests under the same seeds ----
            for fn, kw in ((tf.test_fuse, dict(sim3=False, th=float(rng.choice([2.5, 3.0])), ties=bool(rng.integers(0, 2)))),
                           (tf.test_fuse, dict(sim3=True, th=4.0, ties=bool(rng.integers(0, 2)))),
                           (tf.test_search_by_sim3, dict(n=int(rng.choice([60, 700, 1800])), th=float(rng.choice([7.5, 10.0])))),
                           (tt.test_search_for_triangulation, dict(synth=orb.synth, ori=bool(rng.integers(0, 2)))),
                           (tb.test_last_frame_probes, dict(th=float(rng.choice([7.0, 15.0])))),
                           (tb.test_keyframe_probes, dict()),
                           (tb.test_frustum_probes_and_the_local_map_search,
                            dict(th=float(rng.choice([1.0, 3.0, 5.0])), limit=float(rng.choice([0.5, 0.9])))),
                           (tb.test_fuse_probes, dict(form=str(rng.choice(["pose", "sim3", "sim3_search"]))))):
                fm["calls"] += 1
                try:
                    fn(orb, orbref, **kw)
                except AssertionError as e:
                    if str(e):  # the parity statements of these tests carry a message, their "enough matches" checks do not
                        fm["mismatching_calls"] += 1
                        kw.pop("synth", None)
                        fm["failures"].append(dict(test=fn.__name__, seed=it_seed, what=str(e)[:200], **kw))
        finally:
            np.random.default_rng = real_default_rng
        if time.time() - last_print > 30:
            last_print = time.time()
            print("extraction: %d configurations, %d frames, %d mismatching; scans: %d calls, %d mismatching" %
                  (ex["configurations"], ex["frames"], ex["mismatching_frames"], kn["calls"], kn["mismatching_calls"]), flush=True)
    res = {"what": "tools/parity_sweep.py --seconds %g --seed %d: HIP path against the CPU oracle, bit for bit" % (args.seconds, args.seed),
           "extraction": ex, "hamming_knn2": kn, "windowed_matchers": wm, "search_by_projection": pj,
           "fuse_sim3_triangulation": fm}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({"extraction": {k: v for k, v in ex.items() if k != "failures"}, "failures": ex["failures"][:5],
                      "hamming_knn2": {k: v for k, v in kn.items() if k != "failures"}, "knn_failures": kn["failures"][:5],
                      "windowed_matchers": {k: v for k, v in wm.items() if k != "failures"}, "window_failures": wm["failures"][:5],
                      "search_by_projection": {k: v for k, v in pj.items() if k != "failures"}, "projection_failures": pj["failures"][:5],
                      "fuse_sim3_triangulation": {k: v for k, v in fm.items() if k != "failures"}, "fuse_failures": fm["failures"][:5]}))
    return 1 if (ex["mismatching_frames"] or kn["mismatching_calls"] or wm["mismatching_calls"] or pj["mismatching_calls"]
                 or fm["mismatching_calls"]) else 0


if __name__ == "__main__":
    sys.exit(main())
