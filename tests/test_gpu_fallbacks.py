"""The extractor's rarely taken paths against the oracle: the two-kernel FAST path (score plane in HBM +
separate NMS) that very wide cells use, the NMS sweep over every pixel that replaces the corner list
when it overflows, and the single-lane selections for lists that do not fit the LDS staging areas.
Bar: bit-exact, as everywhere."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _same(rkp, rdesc, dkp, ddesc):
    assert len(rkp) == len(dkp)
    assert np.array_equal(rkp.view(np.uint32), dkp.view(np.uint32))
    assert np.array_equal(rdesc, ddesc)


@pytest.mark.parametrize("case", [(640, 480, 1000, 8, 20, 1, 0), (752, 480, 1500, 8, 12, 1, 3), (320, 240, 500, 6, 20, 0, 1)])
def test_two_kernel_fast_path(orb, orbref, synth, monkeypatch, case):
    w, h, nf, nl, th, score, fi = case
    monkeypatch.setenv("ORBHIP_NO_FUSED_FAST", "1")  # read when the extractor is created
    dev = orb.ORBextractor(nf, 1.2, nl, score, th, device=0, max_width=w, max_height=h, max_batch=3)
    monkeypatch.delenv("ORBHIP_NO_FUSED_FAST")
    ref = orbref.RefExtractor(nf, 1.2, nl, score, th)
    imgs = synth.synth_batch(w, h, 3, first_idx=fi)
    imgs[1] = (imgs[1] // 8 * 8)  # flat: threshold-7 retry
    out = dev.extract_batch(imgs)
    for f in range(3):
        _same(*ref(imgs[f]), *out[f])


def test_very_wide_cells_take_the_two_kernel_path(orb, orbref, synth):
    """1920x300 with 300 features: level-0 cells are wider than the LDS-resident kernel's tile."""
    w, h = 1920, 300
    dev = orb.ORBextractor(300, 1.2, 4, 1, 20, device=0, max_width=w, max_height=h)
    assert max(dev.level_geometry(l)["cellW"] for l in range(4)) > 235
    img = synth.synth_image(w, h, 5)
    _same(*orbref.RefExtractor(300, 1.2, 4, 1, 20)(img), *dev(img))


def test_lists_larger_than_the_lds_staging(orb, orbref, synth):
    """20 000 features at threshold 5 on 2560x1440: level lists above 4096 keys and cells above 1024
    candidates are selected in place in global memory by one lane."""
    w, h, nf = 2560, 1440, 20000
    img = synth.synth_image(w, h, 6)
    dev = orb.ORBextractor(nf, 1.2, 8, 1, 5, device=0, max_width=w, max_height=h)
    assert dev.level_geometry(0)["quota"] > 4096
    rkp, rdesc = orbref.RefExtractor(nf, 1.2, 8, 1, 5)(img)
    _same(rkp, rdesc, *dev(img))
    assert len(rkp) == nf


_CHILD = r"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import orbref
orb = importlib.import_module("orb-slam2_amd")
rng = np.random.default_rng(0)
imgs = [orb.synth.synth_image(640, 480, 2), rng.integers(0, 256, (480, 640), dtype=np.uint8)]
for img in imgs:
    for score in (1, 0):
        rkp, rdesc = orbref.RefExtractor(1000, 1.2, 8, score, 20)(img)
        dkp, ddesc = orb.ORBextractor(1000, 1.2, 8, score, 20, device=0, max_width=640, max_height=480)(img)
        ok = len(rkp) == len(dkp) and np.array_equal(rkp.view(np.uint32), dkp.view(np.uint32)) and np.array_equal(rdesc, ddesc)
        if not ok:
            sys.exit(3)
sys.exit(0)
"""


def test_corner_list_overflow_sweeps_every_pixel():
    """The corner list capacity is read once per process, so the shrunken capacity needs a process of its own."""
    env = dict(os.environ, ORBHIP_FC_CORNER_CAP="16")
    r = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])


def test_cell_selection_tiers(orb, orbref):
    """Cell lists of up to 512 keys are selected in LDS, up to 1024 with the keys left in global memory, above
    that by one lane: a noise frame at three feature counts puts cells in every tier (checked on the
    per-cell candidate counts), in both key widths."""
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    seen = {"lds": 0, "global": 0, "lane": 0}
    for nf in (250, 400, 800):
        for score in (1, 0):
            dev = orb.ORBextractor(nf, 1.2, 4, score, 20, device=0, max_width=640, max_height=480)
            dkp, ddesc = dev(img)
            _same(*orbref.RefExtractor(nf, 1.2, 4, score, 20)(img), dkp, ddesc)
            c = np.concatenate([dev.tap_cells(3, 0, l) for l in range(4)])
            r = np.concatenate([dev.tap_cells(4, 0, l) for l in range(4)])
            sel = c > r
            seen["lds"] += int(((c <= 512) & sel).sum())
            seen["global"] += int(((c > 512) & (c <= 1024) & sel).sum())
            seen["lane"] += int(((c > 1024) & sel).sum())
    assert min(seen.values()) > 10, seen
