"""Cross-checks of the oracle's image primitives against independent implementations that happen to be
in this image (PyTorch's bilinear resampling, SciPy's separable correlation, NumPy's arctan2).  None of
them is OpenCV, so they cannot pin the fixed-point rounding — the tolerance is one grey level — but they
do pin the geometry: sampling positions, border handling and the filter taps."""
import numpy as np
import pytest


def _p(a):
    import ctypes as C
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("sw,sh,dw,dh", [(640, 480, 533, 400), (533, 400, 444, 333), (320, 240, 267, 200), (100, 80, 50, 40)])
def test_resize_follows_half_pixel_bilinear(orbref, synth, sw, sh, dw, dh):
    import torch
    L = orbref.lib()
    src = synth.synth_image(sw, sh, 3)
    dst = np.zeros((dh, dw), np.uint8)
    L.orbref_resize_linear(_p(src), sw, sh, sw, _p(dst), dw, dh, dw)
    ref = torch.nn.functional.interpolate(torch.from_numpy(src.astype(np.float64))[None, None], size=(dh, dw), mode="bilinear",
                                          align_corners=False, antialias=False)[0, 0].numpy()
    err = np.abs(dst.astype(np.float64) - ref)
    assert err.max() <= 1.0 + 1e-6, err.max()       # 11-bit coefficients and two roundings
    assert err.mean() < 0.3, err.mean()              # and close to the nearest integer on average


def test_gaussian_follows_the_float_filter(orbref, synth):
    from scipy.ndimage import correlate1d
    L = orbref.lib()
    w, h = 200, 150
    img = synth.synth_image(w, h, 8)
    pin = np.pad(img, 16, mode="reflect")            # numpy "reflect" = BORDER_REFLECT_101
    pout = np.zeros_like(pin)
    L.orbref_gauss7(_p(pin), w, h, _p(pout))
    x = np.arange(-3, 4)
    k = np.exp(-x * x / 8.0)
    k = np.round(k / k.sum() * 256) / 256           # the 8-bit taps of a sigma-2, 7-tap kernel
    assert list((k * 256).astype(int)) == [18, 34, 49, 55, 49, 34, 18]
    f = correlate1d(correlate1d(pin.astype(np.float64), k, axis=0, mode="nearest"), k, axis=1, mode="nearest")
    ref = np.minimum(np.floor(f + 0.5), 255)
    assert np.array_equal(pout[16:-16, 16:-16], ref[16:-16, 16:-16].astype(np.uint8))  # exact: the filter is integer
    assert np.array_equal(pout[:16], pin[:16])                                        # margin untouched


def test_fast_atan2_tracks_arctan2(orbref):
    L = orbref.lib()
    rng = np.random.default_rng(4)
    worst = 0.0
    for _ in range(5000):
        y, x = (float(v) for v in rng.integers(-60000, 60000, 2))
        if x == 0 and y == 0:
            continue
        a = L.orbref_fast_atan2(y, x)
        t = np.degrees(np.arctan2(y, x)) % 360.0
        d = abs(a - t)
        worst = max(worst, min(d, 360.0 - d))
    assert worst < 0.3, worst  # the polynomial's documented accuracy


def test_fast_corners_against_a_literal_arc_count(orbref):
    """FAST-9/16 by brute force: a pixel is a corner iff some 9 contiguous circle pixels are all brighter than
    v + t or all darker than v - t (no closed form, no shared code with the oracle's measure)."""
    L = orbref.lib()
    rng = np.random.default_rng(5)
    circle = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0),
              (-3, 1), (-2, 2), (-1, 3)]
    h, w, t = 48, 64, 20
    img = (rng.integers(0, 6, (h, w)) * 40 + rng.integers(-5, 6, (h, w))).clip(0, 255).astype(np.uint8)
    want = np.zeros((h, w), bool)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            v = int(img[y, x])
            ring = [int(img[y + dy, x + dx]) for dx, dy in circle]
            br = [p > v + t for p in ring] * 2
            dk = [p < v - t for p in ring] * 2
            want[y, x] = any(all(br[k:k + 9]) or all(dk[k:k + 9]) for k in range(16))
    got = np.zeros((h, w), bool)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            got[y, x] = L.orbref_fast_measure(img.ctypes.data + y * w + x, w) > t
    assert np.array_equal(want, got) and want.sum() > 30
