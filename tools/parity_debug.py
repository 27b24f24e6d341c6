#!/usr/bin/env python3
"""Stage-by-stage comparison of one extractor configuration against the oracle (GPU box).
usage: parity_debug.py w h nf nl sf th score seed [flat]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orbref  # noqa: E402

orb = importlib.import_module("orb-slam2_amd")
w, h, nf, nl = (int(v) for v in sys.argv[1:5])
sf, th, st, seed = float(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
img = orb.synth.synth_image(w, h, seed)
if len(sys.argv) > 9:
    img = (img // 8 * 8).astype(np.uint8)
dev = orb.ORBextractor(nf, sf, nl, st, th, device=0, max_width=w, max_height=h)
ref = orbref.RefExtractor(nf, sf, nl, st, th)
rkp, rdesc = ref(img)
dkp, ddesc = dev(img)
print("fused" if not os.environ.get("ORBHIP_NO_FUSED_FAST") else "unfused", "n ref/dev", len(rkp), len(dkp))
for l in range(nl):
    g = dev.level_geometry(l)
    print("level", l, g)
    rp, dp = ref.plane(l, False), dev.tap_plane(0, 0, l)
    if not np.array_equal(rp, dp):
        ys, xs = np.nonzero(rp != dp)
        print("  pyramid differs at %d px, first (y,x)=(%d,%d) ref %d dev %d; x range %d..%d y range %d..%d" %
              (len(ys), ys[0], xs[0], rp[ys[0], xs[0]], dp[ys[0], xs[0]], xs.min(), xs.max(), ys.min(), ys.max()))
        break
    counts, cand = ref.level_candidates(l)
    dcounts = dev.tap_cells(3, 0, l)
    if not np.array_equal(counts, dcounts):
        bad = np.nonzero(counts != dcounts)[0]
        print("  candidate counts differ in cells", bad[:10], "ref", counts[bad[:10]], "dev", dcounts[bad[:10]])
        break
    if not np.array_equal(ref.level_retain(l), dev.tap_cells(4, 0, l)):
        print("  nToRetain differs\n   counts", counts, "\n   ref   ", ref.level_retain(l), "\n   dev   ", dev.tap_cells(4, 0, l))
        break
    rk, dk = ref.level_keypoints(l), dev.tap_level_keypoints(0, l)
    if not np.array_equal(rk, dk):
        print("  level keypoints differ: ref %d dev %d" % (len(rk), len(dk)))
        break
    rb, db = ref.plane(l, True), dev.tap_plane(1, 0, l)
    if rb is not None and not np.array_equal(rb, db):
        ys, xs = np.nonzero(rb != db)
        print("  blurred plane differs at %d px, first (%d,%d)" % (len(ys), ys[0], xs[0]))
        break
else:
    same_kp = len(rkp) == len(dkp) and np.array_equal(rkp.view(np.uint32), dkp.view(np.uint32))
    print("stages equal; keypoints equal:", same_kp, "descriptors equal:", len(rkp) == len(dkp) and np.array_equal(rdesc, ddesc))
    if len(rkp) == len(dkp) and not same_kp:
        for fld in rkp.dtype.names:
            d = np.nonzero(rkp[fld] != dkp[fld])[0]
            if len(d):
                print("   field", fld, "differs at", d[:5], rkp[fld][d[:5]], dkp[fld][d[:5]])
