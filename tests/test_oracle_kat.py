"""Known-answer tests that pin the CPU oracle (the reference ships none for this path)."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_pattern_table_is_the_reference_table():
    txt = open(os.path.join(HERE, "..", "include", "orb_pattern.inc")).read()
    body = txt[txt.index("*/") + 2:]
    vals = np.array([int(v) for v in body.replace("\n", "").split(",") if v.strip()], np.int8)
    assert vals.size == 1024 and vals.min() == -13 and vals.max() == 12 and int(vals.astype(int).sum()) == -406
    assert list(vals[:4]) == [8, -3, 9, 5] and list(vals[-4:]) == [-1, -6, 0, -11]
    assert hashlib.sha256(vals.tobytes()).hexdigest() == "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"


def test_swar_distance_is_popcount(orbref):
    rng = np.random.default_rng(0)
    rows = [np.zeros(32, np.uint8), np.full(32, 255, np.uint8)]
    for b in (0, 7, 100, 255):
        r = np.zeros(32, np.uint8)
        r[b >> 3] = 1 << (b & 7)
        rows.append(r)
    rows += [rng.integers(0, 256, 32, dtype=np.uint8) for _ in range(50)]
    for a in rows:
        for b in rows:
            assert orbref.hamming(a, b) == int(np.unpackbits(a ^ b).sum())


@pytest.mark.parametrize("nf,quotas", [(1000, [217, 181, 151, 126, 105, 87, 73, 60]),
                                       (2000, [434, 362, 302, 251, 209, 175, 145, 122]),
                                       (4000, [869, 724, 603, 503, 419, 349, 291, 242])])
def test_constructor_tables(orbref, nf, quotas):
    e = orbref.RefExtractor(nf, 1.2, 8, 1, 20)
    assert list(e.table(0)) == quotas
    assert list(e.table(3)) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert list(e.table(4)) == [18, 34, 49, 55, 49, 34, 18]
    sf = e.table(1)
    assert sf[0] == 1.0 and sf[1] == np.float32(1.2) and abs(sf[7] - 1.2 ** 7) < 1e-5
    assert e.GetLevels() == 8 and e.GetScaleFactor() == np.float32(1.2)


@pytest.mark.parametrize("w,h,nf,sizes,grid", [
    (640, 480, 1000, [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)],
     [(5, 6), (5, 6), (4, 5), (4, 5), (3, 4), (3, 4), (3, 4), (3, 4)]),
    (1280, 720, 2000, [(1280, 720), (1067, 600), (889, 500), (741, 417), (617, 347), (514, 289), (429, 241), (357, 201)],
     [(6, 10), (6, 10), (5, 8), (5, 8), (4, 7), (4, 7), (4, 7), (3, 5)]),
])
def test_level_sizes_and_cell_grids(orbref, synth, w, h, nf, sizes, grid):
    e = orbref.RefExtractor(nf, 1.2, 8, 1, 20)
    e(synth.synth_image(w, h, 0))
    for l in range(8):
        assert e.level_size(l) == sizes[l]
        g = e.level_grid(l)
        assert (g["cols"], g["rows"]) == grid[l]
    if (w, h) == (640, 480):
        g = e.level_grid(0)
        assert (g["cellW"], g["cellH"]) == (122, 75)


def test_cvround_is_half_to_even(orbref):
    L = orbref.lib()
    for v, r in [(0.5, 0), (1.5, 2), (2.5, 2), (-0.5, 0), (-1.5, -2), (2.4999, 2), (2.5001, 3), (-2.5, -2)]:
        assert L.orbref_cvround(v) == r


def test_reflect101(orbref):
    L = orbref.lib()
    n = 40
    assert [L.orbref_reflect101(p, n) for p in (-16, -1, 0, 39, 40, 55)] == [16, 1, 0, 39, 38, 23]


def test_fast_atan2_accuracy_and_quadrants(orbref):
    L = orbref.lib()
    rng = np.random.default_rng(1)
    assert L.orbref_fast_atan2(0.0, 0.0) == 0.0
    assert abs(L.orbref_fast_atan2(0.0, 1.0)) < 1e-3
    assert abs(L.orbref_fast_atan2(1.0, 0.0) - 90) < 1e-3
    assert abs(L.orbref_fast_atan2(0.0, -1.0) - 180) < 1e-3
    assert abs(L.orbref_fast_atan2(-1.0, 0.0) - 270) < 1e-3
    for _ in range(5000):
        y, x = (float(v) for v in rng.integers(-3000000, 3000000, 2))
        a = L.orbref_fast_atan2(y, x)
        t = np.degrees(np.arctan2(y, x)) % 360
        assert 0 <= a <= 360 and abs(((a - t + 180) % 360) - 180) < 0.3


def test_resize_known_answers(orbref):
    L = orbref.lib()
    # a constant image stays constant
    src = np.full((50, 60), 137, np.uint8)
    dst = np.zeros((42, 50), np.uint8)
    L.orbref_resize_linear(_p(src), 60, 50, 60, _p(dst), 50, 42, 50)
    assert (dst == 137).all()
    # horizontal ramp 6 -> 5 columns: fx(dx) = (dx + .5) * 1.2 - .5 = .1, 1.3, 2.5, 3.7, 4.9
    src = np.tile(np.array([0, 40, 80, 120, 160, 200], np.uint8), (6, 1))
    dst = np.zeros((5, 5), np.uint8)
    L.orbref_resize_linear(_p(src), 6, 6, 6, _p(dst), 5, 5, 5)
    assert list(dst[0]) == [4, 52, 100, 148, 196]
    assert (dst == dst[0]).all()
    # upscale hits the right-edge clamp: last columns replicate the last source pixel
    src = np.tile(np.array([10, 20, 30], np.uint8), (3, 1))
    dst = np.zeros((3, 7), np.uint8)
    L.orbref_resize_linear(_p(src), 3, 3, 3, _p(dst), 7, 3, 7)
    assert dst[0, 0] == 10 and dst[0, -1] == 30 and (np.diff(dst[0].astype(int)) >= 0).all()


def test_fast_literal_equals_closed_form(orbref):
    """cv::FAST's segment test + cornerScore versus M = max(A, B): corner(t) <=> M > t, score = M - 1,
    strict-maximum NMS with non-corners and out-of-range neighbours counting as 0."""
    L = orbref.lib()
    rng = np.random.default_rng(2)
    for trial in range(6):
        h, w = 40 + trial, 50 + 2 * trial
        img = (rng.integers(0, 8, (h, w)) * 32 + rng.integers(-6, 7, (h, w))).clip(0, 255).astype(np.uint8)
        M = np.zeros((h, w), np.int32)
        for y in range(3, h - 3):
            for x in range(3, w - 3):
                M[y, x] = L.orbref_fast_measure(img.ctypes.data + y * w + x, w)
        for t in (7, 20, 0, 45):
            out = np.zeros((w * h, 3), np.int32)
            n = L.orbref_fast(_p(img), h, w, w, t, _p(out), out.shape[0])
            got = [tuple(r) for r in out[:n]]
            S = np.where(M > t, M - 1, 0)
            exp = []
            for y in range(3, h - 3):
                for x in range(3, w - 3):
                    if M[y, x] > t:
                        nb = S[y - 1:y + 2, x - 1:x + 2].copy()
                        nb[1, 1] = -1
                        # neighbours outside the examined rectangle count as 0
                        if y - 1 < 3: nb[0, :] = np.minimum(nb[0, :], 0)
                        if y + 1 > h - 4: nb[2, :] = np.minimum(nb[2, :], 0)
                        if x - 1 < 3: nb[:, 0] = np.minimum(nb[:, 0], 0)
                        if x + 1 > w - 4: nb[:, 2] = np.minimum(nb[:, 2], 0)
                        if S[y, x] > nb.max():
                            exp.append((x, y, int(S[y, x])))
            assert got == exp, (trial, t)
            assert len(got) > 0 or t == 45


def test_retain_best_observable_prefix(orbref):
    """retainBest then resize(n) keeps the first n elements after nth_element; with ties the kept
    set is a valid top-n and no element beats a kept one."""
    L = orbref.lib()
    rng = np.random.default_rng(3)
    for n, k in [(10, 3), (100, 8), (480, 217), (37, 36)]:
        resp = rng.integers(7, 15, n).astype(np.float32)
        perm = np.zeros(n, np.int32)
        m = L.orbref_nth_prefix(_p(resp), n, k, _p(perm))
        assert m == k and len(set(perm[:k])) == k
        kept = resp[perm[:k]]
        rest = np.delete(resp, perm[:k])
        assert kept.min() >= rest.max()
        full = np.zeros(n, np.int32)
        m2 = L.orbref_retain_best(_p(resp), n, k, _p(full))
        assert m2 >= k and np.array_equal(full[:k], perm[:k])
        assert (resp[full[k:m2]] == kept.min()).all()  # retainBest keeps every tie of the boundary
    perm = np.zeros(5, np.int32)
    assert L.orbref_nth_prefix(_p(np.arange(5, dtype=np.float32)), 5, 7, _p(perm)) == 5
    assert list(perm) == [0, 1, 2, 3, 4]  # nothing moves when size <= n


def test_gaussian_blur_known_answers(orbref):
    L = orbref.lib()
    w, h = 40, 36
    pin = np.full((h + 32, w + 32), 200, np.uint8)
    pout = np.zeros_like(pin)
    L.orbref_gauss7(_p(pin), w, h, _p(pout))
    # taps sum to 257, not 256: a constant c becomes (c * 257^2 + 2^15) >> 16
    assert (pout[16:-16, 16:-16] == min(255, (200 * 257 * 257 + 32768) >> 16)).all()
    assert (pout[:16] == 200).all() and (pout[:, :16] == 200).all()
    pin[:] = 0
    pin[16 + 10, 16 + 12] = 255  # impulse -> outer product of the taps
    L.orbref_gauss7(_p(pin), w, h, _p(pout))
    taps = np.array([18, 34, 49, 55, 49, 34, 18])
    exp = (np.outer(taps, taps) * 255 + 32768) >> 16
    assert np.array_equal(pout[16 + 7:16 + 14, 16 + 9:16 + 16], exp)
    # margin stays unblurred, and the blur reads it: an impulse in the margin leaks into the ROI only
    pin[:] = 0
    pin[15, 20] = 255
    L.orbref_gauss7(_p(pin), w, h, _p(pout))
    assert pout[15, 20] == 255 and pout[16, 20] == ((taps[2] * taps[3] * 255 + 32768) >> 16)


def test_ic_angle_and_descriptor_on_synthetic_patch(orbref):
    L = orbref.lib()
    s = 64
    yy, xx = np.mgrid[0:s, 0:s]
    img = (xx * 3).clip(0, 255).astype(np.uint8)  # brighter to the right -> centroid at +x -> 0 degrees
    c = img.ctypes.data + 32 * s + 32
    a = L.orbref_ic_angle(c, s)
    assert a < 0.5 or a > 359.5
    img2 = np.ascontiguousarray(img.T)  # brighter downwards -> 90 degrees
    a2 = L.orbref_ic_angle(img2.ctypes.data + 32 * s + 32, s)
    assert abs(a2 - 90) < 0.5
    d0 = np.zeros(32, np.uint8)
    d1 = np.zeros(32, np.uint8)
    rng = np.random.default_rng(4)
    tex = rng.integers(0, 256, (s, s), dtype=np.uint8)
    L.orbref_descriptor(tex.ctypes.data + 32 * s + 32, s, 0.0, _p(d0))
    L.orbref_descriptor(tex.ctypes.data + 32 * s + 32, s, 0.0, _p(d1))
    assert np.array_equal(d0, d1) and d0.any()
    # first test pair (8,-3) vs (9,5) at angle 0: bit 0 of byte 0 = I(y-3, x+8) < I(y+5, x+9)
    assert (d0[0] & 1) == int(tex[32 - 3, 32 + 8] < tex[32 + 5, 32 + 9])


def test_knn2_rule_and_ratio_forms(orbref):
    L = orbref.lib()
    q = np.zeros((1, 32), np.uint8)
    t = np.zeros((4, 32), np.uint8)
    t[0, 0] = 0b111  # d = 3
    t[1, 0] = 0b1    # d = 1
    t[2, 0] = 0b10   # d = 1 (tie: first index wins, second distance = 1)
    t[3, 0] = 0b11   # d = 2
    idx, d1, d2 = orbref.knn2(q, t)
    assert (idx[0], d1[0], d2[0]) == (1, 1, 1)
    idx, d1, d2 = orbref.knn2(q, t[:1])
    assert (idx[0], d1[0], d2[0]) == (0, 3, 2147483647)
    idx, d1, d2 = orbref.knn2(q, t[:0])
    assert (idx[0], d1[0], d2[0]) == (-1, 2147483647, 2147483647)
    assert L.orbref_ratio_accept(6, 10, 0.6, 0) == 0 and L.orbref_ratio_accept(6, 10, 0.6, 1) == 1
    assert L.orbref_ratio_accept(5, 10, 0.6, 0) == 1 and L.orbref_ratio_accept(7, 10, 0.6, 2) == 0
    assert L.orbref_ratio_accept(50, 2147483647, 0.6, 0) == 1
    sizes = np.array([5, 50, 0, 4, 49] + [0] * 25, np.int32)
    ind = np.zeros(3, np.int32)
    L.orbref_three_maxima(_p(sizes), 30, _p(ind))
    assert list(ind) == [1, 4, 0]
    sizes[0] = 4
    L.orbref_three_maxima(_p(sizes), 30, _p(ind))
    assert list(ind) == [1, 4, -1]  # third maximum below 10 % of the first is dropped
    assert L.orbref_rot_bin(10.0, 350.0) == 1 and L.orbref_rot_bin(350.0, 10.0) == 11


def test_golden_fixtures(orbref, synth):
    g = np.load(os.path.join(HERE, "golden", "orb_golden.npz"))
    names = sorted(k[:-7] for k in g.files if k.endswith("_params"))
    assert len(names) == 4
    for name in names:
        w, h, nf, nl, th, fi = (int(v) for v in g[name + "_params"])
        img = synth.synth_image(w, h, fi)
        assert hashlib.sha256(img.tobytes()).digest() == g[name + "_img_sha256"].tobytes(), "input generator drifted"
        kp, desc = orbref.RefExtractor(nf, 1.2, nl, 1, th)(img)
        assert np.array_equal(kp.view(np.uint32).reshape(-1, 7), g[name + "_kp"]), name
        assert np.array_equal(desc, g[name + "_desc"]), name
    q, t = synth.synth_descriptors(300, 500, seed=42)
    assert hashlib.sha256(q.tobytes() + t.tobytes()).digest() == g["knn_seed42_sha256"].tobytes()
    assert np.array_equal(np.stack(orbref.knn2(q, t)), g["knn_seed42"])


def test_empty_and_tiny_images(orbref):
    e = orbref.RefExtractor(1000, 1.2, 8, 1, 20)
    kp, desc = e(np.zeros((0, 0), np.uint8))
    assert len(kp) == 0
    with pytest.raises(RuntimeError):
        e(np.zeros((40, 40), np.uint8))
    kp, desc = e(np.full((480, 640), 9, np.uint8))
    assert len(kp) == 0 and desc.shape == (0, 32)


def test_sse2_gaussian_profile_differs_from_the_scalar_one_only_on_exact_ties(orbref):
    """The study profile of the Gaussian column pass (fp32 accumulation, round to nearest even, first w & ~3
    columns) against the scalar profile on planes built to hit ties: the outputs may differ only where the
    integer column sum is an exact tie, there by one grey level towards the even value, and never in the last
    w & 3 columns.  The integer sums are recomputed here from the taps."""
    rng = np.random.default_rng(12)
    taps = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    L = orbref.lib()
    ties = diffs = 0
    for w, h in ((67, 41), (130, 37), (64, 64)):
        # exact ties are planted: a 7-pixel pattern whose tap-weighted sum is 2^15, repeated down a column
        # (constant columns multiply the row sum by 257, which keeps the tie) or along a row (ties in every
        # column of that row, the last w & 3 included)
        plane = rng.integers(0, 256, (h + 32, w + 32)).astype(np.uint8)
        pats = ([222, 234, 88, 14, 216, 98, 101], [121, 88, 226, 16, 194, 81, 188], [18, 201, 141, 159, 86, 90, 149])
        for i, x0 in enumerate((20, 31, 16 + w - 8)):
            plane[:, x0:x0 + 7] = np.array(pats[i], np.uint8)[None, :]
        plane[22:29, :] = np.array(pats[1], np.uint8)[:, None]
        o0 = np.zeros_like(plane)
        o1 = np.zeros_like(plane)
        L.orbref_gauss7(plane.ctypes.data, w, h, o0.ctypes.data)
        L.orbref_set_gauss_profile(1)
        try:
            L.orbref_gauss7(plane.ctypes.data, w, h, o1.ctypes.data)
        finally:
            L.orbref_set_gauss_profile(0)
        p = plane.astype(np.int64)
        rows = sum(taps[k] * p[:, 13 + k:13 + k + w] for k in range(7))          # all rows, ROI columns
        col = sum(taps[k] * rows[13 + k:13 + k + h] for k in range(7))            # ROI rows
        assert np.array_equal(o0[16:16 + h, 16:16 + w], np.minimum((col + 32768) >> 16, 255))
        tie = (col % 65536) == 32768
        tie[:, (w & ~3):] = False
        even = np.minimum(np.where(tie & (((col + 32768) >> 16) % 2 == 1), ((col + 32768) >> 16) - 1, (col + 32768) >> 16), 255)
        assert np.array_equal(o1[16:16 + h, 16:16 + w], even)
        assert np.array_equal(o1[:16], plane[:16]) and np.array_equal(o1[:, :16], plane[:, :16])
        ties += int(tie.sum())
        diffs += int((o0 != o1).sum())
    assert ties > 20 and 0 < diffs <= ties
