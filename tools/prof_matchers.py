#!/usr/bin/env python3
"""The matcher legs of bench.py alone, for rocprofv3 passes over the matcher kernels (k_knn2, k_knn2_fold,
k_knn2_merge, k_area_*, k_resolve_init, k_probe_topk, k_resolve_probe, k_frame_grid, k_bow_*):

  rocprofv3 --kernel-trace --stats -d <dir> -- python3 tools/prof_matchers.py [knn|windowed|bow ...]
"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    which = set(sys.argv[1:]) or {"knn", "windowed", "bow"}
    orb = importlib.import_module("orb-slam2_amd")
    import torch
    dev = 0
    torch.cuda.set_device(dev)
    orbref = bench.oracle()
    m = orb.ORBmatcher(0.6, True, device=dev)
    res = {}
    if "knn" in which:
        ms, _, q, t, o = bench.knn_pairs_leg(orb, torch, dev, m, reps=3)
        res["knn2_4096_pairs_ms"] = ms
        del q, t, o
        g = torch.Generator(device="cuda")
        g.manual_seed(2)
        q2 = torch.randint(0, 256, (4000, 32), dtype=torch.uint8, device="cuda", generator=g)
        t2 = torch.randint(0, 256, (1250000, 32), dtype=torch.uint8, device="cuda", generator=g)
        o2 = torch.zeros((3, 4000), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for _ in range(3):
            m.knn2_device(q2.data_ptr(), 4000, 0, t2.data_ptr(), 1250000, 0, 1, 0, o2[0].data_ptr(), o2[1].data_ptr(),
                          o2[2].data_ptr())
            m.synchronize()
        res["knn2_map_shard_ms"] = m.last_elapsed_ms()
    if "windowed" in which:
        res["windowed"] = bench.windowed_leg(orb, torch, dev, orbref)
    if "bow" in which:
        res["bow"] = bench.bow_leg(orb, torch, dev, orbref, m)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
