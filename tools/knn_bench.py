#!/usr/bin/env python3
"""Quick timing of the Hamming kNN kernel on the two benchmark shapes (4096 pairs of 2000 x 2000;
4000 x 1.25 M map shard), with the sustained v_bcnt rate of the device as the denominator and a parity
spot check against the oracle.  Development tool; bench.py reports the same legs."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import torch
    import orbref
    orb = importlib.import_module("orb-slam2_amd")
    m = orb.ORBmatcher(0.6, True)
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    r_bcnt = m.valu_rate(0)
    print("sustained v_bcnt lane-ops/s %.4g" % r_bcnt)
    for name, pairs, nq, nt in (("pairs4096", 4096, 2000, 2000), ("map_shard", 1, 4000, 1250000)):
        q = torch.randint(0, 256, (pairs, nq, 32), dtype=torch.uint8, device="cuda", generator=g)
        t = torch.randint(0, 256, (pairs, nt, 32), dtype=torch.uint8, device="cuda", generator=g)
        o = torch.zeros((3, pairs, nq), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ms = []
        for _ in range(6):
            m.knn2_device(q.data_ptr(), nq, nq * 32, t.data_ptr(), nt, nt * 32, pairs, 0, o[0].data_ptr(), o[1].data_ptr(),
                          o[2].data_ptr())
            m.synchronize()
            ms.append(m.last_elapsed_ms())
        best = float(np.median(ms[1:]))
        rate = pairs * nq * nt / (best * 1e-3)
        k = 64 if pairs == 1 else nq
        ridx, rd1, rd2 = orbref.knn2(q[0, :k].cpu().numpy(), t[0].cpu().numpy())
        oo = o[:, 0, :k].cpu().numpy()
        ok = np.array_equal(oo[0], ridx) and np.array_equal(oo[1], rd1) and np.array_equal(oo[2], rd2)
        print("%s: %.3f ms  %.4g comparisons/s  %.1f%% of sustained v_bcnt  parity=%s" % (name, best, rate, 100 * rate * 8 / r_bcnt, ok))


if __name__ == "__main__":
    main()
