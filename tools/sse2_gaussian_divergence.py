#!/usr/bin/env python3
"""How far would an SSE2 build of OpenCV 2.4 move the results from the oracle's scalar profile?

The one known difference between the two is the column pass of the 8-bit Gaussian (the oracle's header and
gauss7_roi in oracle/orb_ref.cpp): fp32 accumulation with round-to-nearest-even for the first w & ~3 columns
instead of (sum + 2^15) >> 16.  This script runs the oracle under both profiles over the seeded synthetic frames
of the benchmark configurations and counts blurred pixels, descriptors and descriptor bits that differ.  CPU
only; nothing of the product is involved.

  python3 tools/sse2_gaussian_divergence.py [--frames N]
"""
import argparse
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orbref  # noqa: E402


def run(ext, img, profile):
    orbref.lib().orbref_set_gauss_profile(profile)
    try:
        kp, desc = ext(img)
        planes = [ext.plane(l, blurred=True).copy() for l in range(ext.GetLevels())]
    finally:
        orbref.lib().orbref_set_gauss_profile(0)
    return kp, desc, planes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    args = ap.parse_args()
    synth = importlib.import_module("orb-slam2_amd.synth")
    report = {}
    for w, h, nf in ((640, 480, 1000), (1280, 720, 2000)):
        ext = orbref.RefExtractor(nf, 1.2, 8, 1, 20)
        tot = {"frames": 0, "blurred_pixels": 0, "blurred_pixels_differing": 0, "keypoints": 0, "keypoint_records_differing": 0,
               "descriptors_differing": 0, "descriptor_bits": 0, "descriptor_bits_differing": 0, "max_pixel_difference": 0}
        for f in range(args.frames):
            img = synth.synth_image(w, h, f)
            kp0, d0, p0 = run(ext, img, 0)
            kp1, d1, p1 = run(ext, img, 1)
            tot["frames"] += 1
            for a, b in zip(p0, p1):
                roi_a, roi_b = a[16:-16, 16:-16].astype(np.int16), b[16:-16, 16:-16].astype(np.int16)
                tot["blurred_pixels"] += roi_a.size
                tot["blurred_pixels_differing"] += int((roi_a != roi_b).sum())
                tot["max_pixel_difference"] = max(tot["max_pixel_difference"], int(np.abs(roi_a - roi_b).max()))
            assert len(kp0) == len(kp1)
            tot["keypoints"] += len(kp0)
            tot["keypoint_records_differing"] += int((kp0.view(np.uint32).reshape(len(kp0), -1)
                                                      != kp1.view(np.uint32).reshape(len(kp1), -1)).any(1).sum())
            x = np.unpackbits(d0 ^ d1, axis=1).sum(1)
            tot["descriptors_differing"] += int((x > 0).sum())
            tot["descriptor_bits"] += d0.size * 8
            tot["descriptor_bits_differing"] += int(x.sum())
        tot["blurred_pixel_share"] = tot["blurred_pixels_differing"] / max(tot["blurred_pixels"], 1)
        tot["descriptor_bit_share"] = tot["descriptor_bits_differing"] / max(tot["descriptor_bits"], 1)
        report["%dx%d/%d" % (w, h, nf)] = tot
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
