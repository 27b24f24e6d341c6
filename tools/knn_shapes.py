#!/usr/bin/env python3
"""Development tool: duration of the Hamming kNN scan on one shape, call after call, with the sustained
v_bcnt rate of the device measured before and after (the chip's clock moves with load, so a rate is only
comparable with a denominator taken next to it)."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

orb = importlib.import_module("orb-slam2_amd")
m = orb.ORBmatcher(0.6, True)
g = torch.Generator(device="cuda")
g.manual_seed(1)
nq, nt = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4000, 1250000)
q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device="cuda", generator=g)
t = torch.randint(0, 256, (nt, 32), dtype=torch.uint8, device="cuda", generator=g)
o = torch.zeros((3, nq), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for rnd in range(3):
    r0 = m.valu_rate(0)
    ms = []
    for _ in range(20):
        m.knn2_device(q.data_ptr(), nq, 0, t.data_ptr(), nt, 0, 1, 0, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
        m.synchronize()
        ms.append(m.last_elapsed_ms())
    r1 = m.valu_rate(0)
    med = float(np.median(ms))
    print("round %d: v_bcnt before %.4g after %.4g; ms min %.3f median %.3f max %.3f; %.1f%% of the mean rate"
          % (rnd, r0, r1, min(ms), med, max(ms), 100 * nq * nt / (med * 1e-3) * 8 / (0.5 * (r0 + r1))))
    print("   ", " ".join("%.3f" % v for v in ms))
