#!/usr/bin/env python3
"""Timing of the three acceptance rules of the projection matchers on a dense case (256 frames of ~2000 keypoints,
one probe per keypoint of the other frame, radius r * scale).  Development tool."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    orb = importlib.import_module("orb-slam2_amd")
    import torch
    dev, b, w, h, nf = 0, 256, 1280, 720, 2000
    wl = bench.ExtractWorkload(orb, torch, dev, w, h, nf, b, first_idx=200000)
    wl.step()
    wl.sync()
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    kp1 = wl.d_kp.view(torch.uint8).view(b, nf, 28).clone()
    kp2 = kp1.clone()
    kp2.view(torch.float32).view(b, nf, 7)[:, :, 0:2] += torch.rand((b, nf, 2), device="cuda", generator=g) * 16 - 8
    d1 = wl.d_desc.view(b, nf, 32).clone()
    flips = torch.randint(0, 256, (b, nf, 32), dtype=torch.uint8, device="cuda", generator=g)
    for _ in range(4):
        flips &= torch.randint(0, 256, (b, nf, 32), dtype=torch.uint8, device="cuda", generator=g)
    d2 = d1 ^ flips
    n = wl.d_n.clone()
    m = orb.ORBmatcher(0.9, True, device=dev)
    F2 = orb.FrameBatch(m, kp2, d2, n, [0, 0, w, h])
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    k1f = kp1.view(torch.float32).view(b, nf, 7)
    octv = kp1.view(torch.int32).view(b, nf, 7)[:, :, 5]
    for radius in (15.0, 60.0):
        pr = torch.zeros((b, nf, 8), dtype=torch.float32, device="cuda")
        pri = pr.view(torch.int32)
        pr[:, :, 0:2] = k1f[:, :, 0:2]
        pr[:, :, 2] = torch.from_numpy(sf * np.float32(radius)).cuda()[octv.long()]
        pri[:, :, 3] = octv - 1
        pri[:, :, 4] = octv + 1
        pr[:, :, 5] = k1f[:, :, 3]
        pri[:, :, 6] = 1
        probes = pr.view(torch.uint8).view(b, nf, 32)
        for name, rule in (("best", 2), ("ratio", 1), ("local_map", 0)):
            ts = []
            for _ in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                nm, mo = m.SearchByProjection(F2, probes, d1, n, rule, th_dist=100)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            print("radius %g rule %-9s %.3f ms per %d frames, %.1f matches/frame, checksum %d" %
                  (radius, name, np.median(ts[1:]) * 1e3, b, float(nm.float().mean()), int(mo.long().sum())))


if __name__ == "__main__":
    main()
