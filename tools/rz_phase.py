#!/usr/bin/env python3
"""Phase shares of k_resize (wave 0 of every workgroup) from a library built with -DORBHIP_RZ_PHASE_TIMING (developer
instrumentation; ORBHIP_LIBRARY names that build)."""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

orb = importlib.import_module("orb-slam2_amd")
import torch  # noqa: E402

w, h, nf = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (640, 480, 1000)
wl = bench.ExtractWorkload(orb, torch, 0, w, h, nf, 1024, 0)
L = orb._lib.lib()
out = (C.c_ulonglong * 8)()
wl.step(); wl.sync()
L.orbhip_debug_rz_phase(out)
for _ in range(5):
    wl.step()
wl.sync()
L.orbhip_debug_rz_phase(out)
names = ["tile rectangle (scalar loads)", "staging loads + LDS stores", "column tables", "barrier", "rows", "-", "-", "-"]
tot = sum(out)
for n, v in zip(names, out):
    if v:
        print("%-32s %6.2f %%  %8.0f cycles per workgroup" % (n, 100.0 * v / tot, v / (5 * 1024 * 68.0)))
