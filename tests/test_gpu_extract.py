"""Parity of the HIP extractor against the CPU oracle, through the C ABI (ctypes), stage by stage
and end to end.  Bar: keypoint coordinates, octaves, responses, sizes and descriptor bytes
bit-exact; angles bit-exact as well (the north-star tolerance is 1e-4 degrees; the device uses the
same fp32 polynomial without contraction, so equality is what is asserted and the tolerance is
only reported)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    # (width, height, nfeatures, nlevels, fastTh, frame_idx)
    (640, 480, 1000, 8, 20, 0),
    (640, 480, 1000, 8, 20, 7),
    (320, 240, 500, 6, 20, 1),
    (1280, 720, 2000, 8, 20, 2),
    (752, 480, 1500, 8, 12, 3),
    (500, 375, 800, 5, 30, 4),
    (1920, 1080, 4000, 8, 20, 9),
]


def _make(orb, orbref, w, h, nf, nl, th, **kw):
    dev = orb.ORBextractor(nf, 1.2, nl, 1, th, device=0, max_width=w, max_height=h, max_batch=kw.get("max_batch", 1))
    ref = orbref.RefExtractor(nf, 1.2, nl, 1, th)
    return dev, ref


@pytest.mark.parametrize("case", CASES)
def test_tables_and_geometry(orb, orbref, synth, case):
    w, h, nf, nl, th, fi = case
    dev, ref = _make(orb, orbref, w, h, nf, nl, th)
    for which in range(5):
        assert np.array_equal(dev.table(which), ref.table(which)), which
    assert dev.GetLevels() == ref.GetLevels()
    assert dev.GetScaleFactor() == ref.GetScaleFactor()
    ref(synth.synth_image(w, h, fi))
    for l in range(nl):
        g = dev.level_geometry(l)
        rg = ref.level_grid(l)
        assert (g["w"], g["h"]) == ref.level_size(l)
        for k in ("cols", "rows", "cellW", "cellH", "nfeaturesCell"):
            assert g[k] == rg[k], (l, k)


@pytest.mark.parametrize("case", CASES)
def test_stages(orb, orbref, synth, case):
    w, h, nf, nl, th, fi = case
    dev, ref = _make(orb, orbref, w, h, nf, nl, th)
    img = synth.synth_image(w, h, fi)
    rkp, rdesc = ref(img)
    dkp, ddesc = dev(img)
    for l in range(nl):
        rp = ref.plane(l, blurred=False)
        dp = dev.tap_plane(0, 0, l)
        assert np.array_equal(rp, dp), "pyramid level %d: %d px differ" % (l, int((rp != dp).sum()))
    for l in range(nl):
        counts, cand = ref.level_candidates(l)
        dcounts = dev.tap_cells(3, 0, l)
        assert np.array_equal(counts, dcounts), "level %d candidate counts" % l
        dcand = dev.tap_candidates(0, l)
        # after selection a cell's device list is a permutation of its raster-ordered candidates
        o = 0
        for c, n in enumerate(counts):
            a = sorted(map(tuple, cand[o:o + n]))
            b = sorted(map(tuple, dcand[o:o + n]))
            assert a == b, "level %d cell %d candidates" % (l, c)
            o += n
        retain = ref.level_retain(l)
        assert np.array_equal(retain, dev.tap_cells(4, 0, l)), "level %d nToRetain" % l
        o = 0
        for c, n in enumerate(counts):  # a cell that keeps everything is never permuted: raster order
            if n <= retain[c]:
                assert np.array_equal(cand[o:o + n], dcand[o:o + n]), "level %d cell %d raster order" % (l, c)
            o += n
        assert np.array_equal(ref.level_keypoints(l), dev.tap_level_keypoints(0, l)), "level %d selection" % l
    for l in range(nl):
        rb = ref.plane(l, blurred=True)
        if rb is None:
            continue
        db = dev.tap_plane(1, 0, l)
        assert np.array_equal(rb, db), "blurred level %d: %d px differ" % (l, int((rb != db).sum()))
    assert len(rkp) == len(dkp)


def _assert_same(rkp, rdesc, dkp, ddesc):
    assert len(rkp) == len(dkp)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(rkp[f], dkp[f]), f
    dang = np.abs(rkp["angle"].astype(np.float64) - dkp["angle"].astype(np.float64))
    assert dang.max(initial=0.0) <= 1e-4, "orientation differs by %g deg" % dang.max()
    assert np.array_equal(rkp["angle"].view(np.uint32), dkp["angle"].view(np.uint32))
    flips = int(np.unpackbits(rdesc ^ ddesc).sum())
    assert flips == 0, "%d descriptor bits differ" % flips


@pytest.mark.parametrize("case", CASES)
def test_end_to_end(orb, orbref, synth, case):
    w, h, nf, nl, th, fi = case
    dev, ref = _make(orb, orbref, w, h, nf, nl, th)
    img = synth.synth_image(w, h, fi)
    rkp, rdesc = ref(img)
    dkp, ddesc = dev(img)
    _assert_same(rkp, rdesc, dkp, ddesc)
    assert len(rkp) > 0


def test_one_frame_per_call_replays_and_recaptures(orb, orbref, synth):
    """One frame per call goes through a launch sequence captured once per image size and replayed: repeated
    calls with different images, a change of image size, a batched call in between that reallocates the
    handle's buffers, and a change of the Gaussian profile must each give the oracle's result."""
    dev = orb.ORBextractor(800, 1.2, 8, 1, 20, device=0, max_width=752, max_height=480, max_batch=4)
    ref = orbref.RefExtractor(800, 1.2, 8, 1, 20)

    def check(img):
        rkp, rdesc = ref(img)
        dkp, ddesc = dev(img)
        _assert_same(rkp, rdesc, dkp, ddesc)

    for i in range(4):
        check(synth.synth_image(640, 480, 40 + i))
        # the device time of the call just made, replayed or not: some tens of microseconds to a few milliseconds
        assert 0.01 < dev.last_elapsed_ms() < 20.0, dev.last_elapsed_ms()
    check(synth.synth_image(752, 480, 50))  # new size: new capture
    check(synth.synth_image(640, 480, 51))
    imgs = synth.synth_batch(640, 480, 4, first_idx=60)  # larger arenas: the captured addresses are stale
    outs = dev.extract_batch(imgs)
    for img, (dkp, ddesc) in zip(imgs, outs):
        rkp, rdesc = ref(img)
        _assert_same(rkp, rdesc, dkp, ddesc)
    for i in range(3):
        check(synth.synth_image(640, 480, 70 + i))
    dev.set_gaussian_profile(dev.GAUSS_SSE2)
    orbref.lib().orbref_set_gauss_profile(1)
    try:
        check(synth.synth_image(640, 480, 80))
    finally:
        orbref.lib().orbref_set_gauss_profile(0)
        dev.set_gaussian_profile(dev.GAUSS_SCALAR)
    check(synth.synth_image(640, 480, 81))


def test_batch_matches_single(orb, orbref, synth):
    w, h, nf = 640, 480, 1000
    dev, ref = _make(orb, orbref, w, h, nf, 8, 20, max_batch=4)
    imgs = synth.synth_batch(w, h, 6, first_idx=20)  # 6 frames through chunks of 4
    out = dev.extract_batch(imgs)
    assert len(out) == 6
    for f in range(6):
        rkp, rdesc = ref(imgs[f])
        _assert_same(rkp, rdesc, out[f][0], out[f][1])


@pytest.mark.parametrize("score", [1, 0])
def test_large_batch_frame_groups(orb, orbref, synth, score):
    """One chunk of 19 frames: two full groups of eight frames (whose workgroups the kernels hand to the
    eight dies one frame each) and a tail of three, every frame checked against the oracle."""
    w, h, nf, nb = 376, 240, 600, 19
    dev = orb.ORBextractor(nf, 1.2, 6, score, 20, device=0, max_width=w, max_height=h, max_batch=nb)
    ref = orbref.RefExtractor(nf, 1.2, 6, score, 20)
    imgs = synth.synth_batch(w, h, nb, first_idx=40)
    imgs[5] = 90                               # a blank frame inside a group
    imgs[17] = (imgs[17] // 8 * 8)             # a flat one in the tail: threshold-7 fallback
    out = dev.extract_batch(imgs)
    assert len(out) == nb
    for f in range(nb):
        rkp, rdesc = ref(imgs[f])
        _assert_same(rkp, rdesc, out[f][0], out[f][1])


@pytest.mark.parametrize("w,h,nf,nb", [(1280, 720, 2000, 16), (1920, 1080, 4000, 8)])
def test_full_size_frames_through_one_chunk(orb, orbref, synth, w, h, nf, nb):
    """The two larger configurations as batches: 16 frames of 1280x720 / 2000 features and 8 of 1920x1080 / 4000
    through ONE chunk of the batched entry point (two full groups of eight frames / one), every frame against the
    oracle."""
    dev = orb.ORBextractor(nf, 1.2, 8, 1, 20, device=0, max_width=w, max_height=h, max_batch=nb)
    ref = orbref.RefExtractor(nf, 1.2, 8, 1, 20)
    imgs = synth.synth_batch(w, h, nb, first_idx=700)
    out = dev.extract_batch(imgs)
    assert len(out) == nb
    total = 0
    for f in range(nb):
        rkp, rdesc = ref(imgs[f])
        _assert_same(rkp, rdesc, out[f][0], out[f][1])
        total += len(rkp)
    assert total > nb * nf * 0.9


def test_strided_input_and_reuse(orb, orbref, synth):
    """A view with a row stride larger than its width, then a different size on the same handle."""
    dev, ref = _make(orb, orbref, 640, 480, 1000, 8, 20)
    big = synth.synth_image(800, 600, 30)
    view = big[50:530, 100:740]
    assert view.strides[0] == 800
    rkp, rdesc = ref(np.ascontiguousarray(view))
    dkp, ddesc = dev(view)
    _assert_same(rkp, rdesc, dkp, ddesc)
    img2 = synth.synth_image(512, 384, 31)
    rkp, rdesc = ref(img2)
    dkp, ddesc = dev(img2)
    _assert_same(rkp, rdesc, dkp, ddesc)


def test_sparse_image_uses_threshold_7(orb, orbref):
    """Mostly flat frame with a few weak corners: cells fall back to FAST threshold 7
    (src/ORBextractor.cc:609-614) and many cells stay empty (quota redistribution)."""
    rng = np.random.default_rng(5)
    img = np.full((480, 640), 100, np.uint8)
    for _ in range(60):
        x, y = int(rng.integers(30, 600)), int(rng.integers(30, 440))
        img[y:y + 9, x:x + 9] = 100 + int(rng.integers(9, 19))
    for _ in range(25):
        x, y = int(rng.integers(30, 600)), int(rng.integers(30, 440))
        img[y:y + 12, x:x + 12] = 200
    dev, ref = _make(orb, orbref, 640, 480, 1000, 8, 20)
    rkp, rdesc = ref(img)
    dkp, ddesc = dev(img)
    assert len(rkp) > 0
    assert (rkp["response"] < 20).any()
    _assert_same(rkp, rdesc, dkp, ddesc)


def test_blank_image_gives_no_keypoints(orb, orbref):
    img = np.full((480, 640), 77, np.uint8)
    dev, ref = _make(orb, orbref, 640, 480, 1000, 8, 20)
    rkp, _ = ref(img)
    dkp, ddesc = dev(img)
    assert len(rkp) == 0 and len(dkp) == 0 and ddesc.shape == (0, 32)


def test_empty_image_and_errors(orb):
    dev = orb.ORBextractor(1000, 1.2, 8, 1, 20, device=0, max_width=640, max_height=480)
    assert dev(np.zeros((0, 0), np.uint8)) == (None, None)
    with pytest.raises(AssertionError):
        dev(np.zeros((480, 640, 3), np.uint8))
    with pytest.raises(orb.OrbhipError) as ei:
        dev(np.zeros((40, 40), np.uint8))  # no interior inside the 16-px border at the top levels
    assert ei.value.code == -5
    with pytest.raises(orb.OrbhipError):
        orb.ORBextractor(0, 1.2, 8, 1, 20)


HARRIS_CASES = [(640, 480, 1000, 8, 20, 0), (1280, 720, 2000, 8, 20, 2), (320, 240, 500, 6, 12, 5)]


@pytest.mark.parametrize("case", HARRIS_CASES)
def test_harris_score(orb, orbref, synth, case):
    """scoreType = HARRIS_SCORE (src/ORBextractor.cc:616-620): the float Harris response replaces the FAST
    score before the quota / retainBest selection.  Responses are compared as IEEE bits."""
    w, h, nf, nl, th, fi = case
    dev = orb.ORBextractor(nf, 1.2, nl, orb.ORBextractor.HARRIS_SCORE, th, device=0, max_width=w, max_height=h)
    ref = orbref.RefExtractor(nf, 1.2, nl, 0, th)
    img = synth.synth_image(w, h, fi)
    rkp, rdesc = ref(img)
    dkp, ddesc = dev(img)
    for l in range(nl):
        counts, cand = ref.level_candidates(l)
        assert np.array_equal(counts, dev.tap_cells(3, 0, l)), "level %d candidate counts" % l
        dcand = dev.tap_candidates(0, l)
        o = 0
        for c, n in enumerate(counts):  # same candidates with the same response bits, as a set per cell
            assert sorted(map(tuple, cand[o:o + n])) == sorted(map(tuple, dcand[o:o + n])), "level %d cell %d" % (l, c)
            o += n
        assert np.array_equal(ref.level_retain(l), dev.tap_cells(4, 0, l)), "level %d nToRetain" % l
        assert np.array_equal(ref.level_keypoints(l), dev.tap_level_keypoints(0, l)), "level %d selection" % l
    _assert_same(rkp, rdesc, dkp, ddesc)
    assert len(rkp) > nf // 2
    assert (rkp["response"] != np.round(rkp["response"])).any()  # these are Harris floats, not FAST integers


def test_harris_batch_and_mode_switch(orb, orbref, synth):
    """A batch through the Harris path, then a FAST-score extractor in the same process."""
    w, h, nf = 640, 480, 800
    imgs = np.stack([synth.synth_image(w, h, 20 + i) for i in range(5)])
    dev = orb.ORBextractor(nf, 1.2, 8, 0, 20, device=0, max_width=w, max_height=h, max_batch=2)
    ref = orbref.RefExtractor(nf, 1.2, 8, 0, 20)
    out = dev.extract_batch(imgs)
    for i in range(5):
        rkp, rdesc = ref(imgs[i])
        _assert_same(rkp, rdesc, out[i][0], out[i][1])
    devf = orb.ORBextractor(nf, 1.2, 8, 1, 20, device=0, max_width=w, max_height=h)
    rkp, rdesc = orbref.RefExtractor(nf, 1.2, 8, 1, 20)(imgs[0])
    dkp, ddesc = devf(imgs[0])
    _assert_same(rkp, rdesc, dkp, ddesc)


# (width, height, nfeatures, nlevels, scaleFactor, fastTh, scoreType, frame_idx, flattened) — extreme aspect
# ratios, one-level pyramids, scale 2.0, cells wider than the LDS-resident FAST kernel takes, a last column
# whose cell image is 1..6 px wide (visited by the reference, nothing examined: it closes in the first quota
# pass with zero keys, unlike a column the reference skips)
ODD_GEOMETRY_CASES = [
    (113, 624, 232, 3, 1.2, 24, 0, 9807, False),
    (962, 156, 369, 3, 1.1, 29, 0, 3856, False),
    (306, 781, 225, 3, 1.1, 31, 1, 7332, False),
    (369, 723, 2131, 2, 2.0, 11, 1, 3906, False),
    (216, 614, 356, 6, 1.1, 5, 1, 1951, True),
    (817, 290, 2530, 8, 1.2, 29, 0, 756, True),
    (948, 198, 653, 2, 1.2, 5, 1, 9394, True),
    (867, 584, 2602, 1, 2.0, 42, 1, 7515, False),
    (380, 140, 2327, 7, 1.2, 11, 0, 3751, False),
]


@pytest.mark.parametrize("case", ODD_GEOMETRY_CASES)
def test_odd_geometries(orb, orbref, synth, case):
    w, h, nf, nl, sf, th, st, fi, flat = case
    img = synth.synth_image(w, h, fi)
    if flat:
        img = (img // 8 * 8).astype(np.uint8)
    dev = orb.ORBextractor(nf, sf, nl, st, th, device=0, max_width=w, max_height=h)
    ref = orbref.RefExtractor(nf, sf, nl, st, th)
    rkp, rdesc = ref(img)
    dkp, ddesc = dev(img)
    for l in range(nl):
        assert np.array_equal(ref.level_candidates(l)[0], dev.tap_cells(3, 0, l)), "level %d candidate counts" % l
        assert np.array_equal(ref.level_retain(l), dev.tap_cells(4, 0, l)), "level %d nToRetain" % l
    _assert_same(rkp, rdesc, dkp, ddesc)
    assert len(rkp) > 0


def test_cells_past_the_border_are_rejected(orb, orbref, synth):
    """ceil() cell sizes times the cell count can carry a cell that is not in the last row past maxBorderY
    (src/ORBextractor.cc:541-545, :564); a corner found there has its 31x31 patch outside the image, which the
    reference reads unchecked.  The device refuses the geometry; the oracle reports the first such corner."""
    with pytest.raises(orb.OrbhipError) as ei:
        orb.ORBextractor(858, 1.3, 1, 1, 43, device=0, max_width=807, max_height=139)(synth.synth_image(807, 139, 7533))
    assert ei.value.code == -5 and "border" in str(ei.value)
    with pytest.raises(RuntimeError):
        orbref.RefExtractor(858, 1.3, 1, 1, 43)(synth.synth_image(807, 139, 7533))


@pytest.mark.parametrize("kind", ["noise", "noise_binary", "checker1", "checker3", "dots7", "ramp"])
@pytest.mark.parametrize("nf,th,score", [(3000, 5, 1), (1000, 20, 0)])
def test_dense_corner_images(orb, orbref, kind, nf, th, score):
    """Corner densities far above natural images (survivor lists, corner lists and cell lists at their
    fullest; heavy score ties in the selection)."""
    w, h = 640, 480
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:h, 0:w]
    img = {
        "noise": lambda: rng.integers(0, 256, (h, w), dtype=np.uint8),
        "noise_binary": lambda: (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8),
        "checker1": lambda: (((xx + yy) & 1) * 255).astype(np.uint8),
        "checker3": lambda: ((((xx // 3) + (yy // 3)) & 1) * 255).astype(np.uint8),
        "dots7": lambda: np.where((xx % 7 == 3) & (yy % 7 == 3), 255, 0).astype(np.uint8),
        "ramp": lambda: ((xx * 3 + yy * 5) % 256).astype(np.uint8),
    }[kind]()
    dev = orb.ORBextractor(nf, 1.2, 8, score, th, device=0, max_width=w, max_height=h)
    ref = orbref.RefExtractor(nf, 1.2, 8, score, th)
    rkp, rdesc = ref(img)
    dkp, ddesc = dev(img)
    _assert_same(rkp, rdesc, dkp, ddesc)
    assert len(rkp) > 500


def test_sse2_gaussian_profile(orb, orbref, synth):
    """ORBHIP_GAUSS_SSE2: blurred planes and descriptors equal the oracle's SSE2 study profile (ties of the column
    pass to even in the first w & ~3 columns); on an image of few grey levels the two profiles really differ."""
    img = synth.synth_image(322, 240, 9)
    # 7-pixel patterns whose tap-weighted sum is 2^15, repeated down columns and along rows: exact ties of the
    # column pass at level 0, also in the last w & 3 columns where both profiles round up
    pats = ([222, 234, 88, 14, 216, 98, 101], [121, 88, 226, 16, 194, 81, 188], [18, 201, 141, 159, 86, 90, 149])
    for i, x0 in enumerate(range(12, 300, 37)):
        img[i * 9:i * 9 + 150, x0:x0 + 7] = np.array(pats[i % 3], np.uint8)[None, :]
    img[200:207, 30:] = np.array(pats[1], np.uint8)[:, None]
    ext = orb.ORBextractor(500, 1.2, 8, 1, 20, max_width=322, max_height=240)
    ref = orbref.RefExtractor(500, 1.2, 8, 1, 20)
    kp0, d0 = ext(img)
    ext.set_gaussian_profile(ext.GAUSS_SSE2)
    kp1, d1 = ext(img)
    dev_planes = [ext.tap_plane(1, 0, l) for l in range(8)]
    orbref.lib().orbref_set_gauss_profile(1)
    try:
        rkp, rd = ref(img)
        ref_planes = [ref.plane(l, blurred=True).copy() for l in range(8)]
    finally:
        orbref.lib().orbref_set_gauss_profile(0)
    rkp0, rd0 = ref(img)
    scalar_planes = [ref.plane(l, blurred=True).copy() for l in range(8)]
    assert np.array_equal(kp0.view(np.uint32), rkp0.view(np.uint32)) and np.array_equal(d0, rd0)
    assert np.array_equal(kp1.view(np.uint32), rkp.view(np.uint32)) and np.array_equal(d1, rd)
    for a, b in zip(dev_planes, ref_planes):
        assert np.array_equal(a, b)
    assert sum(int((a != b).sum()) for a, b in zip(ref_planes, scalar_planes)) > 10
    ext.set_gaussian_profile(ext.GAUSS_SCALAR)
    kp2, d2 = ext(img)
    assert np.array_equal(d2, d0)
    with pytest.raises(orb.OrbhipError):
        ext.set_gaussian_profile(7)


def test_batch_from_page_locked_host_memory(orb, synth):
    """orbhip_extract_batch copies packed frames that lie in page-locked host memory straight to the device and
    stages pageable ones: both must give the same keypoints and descriptors, with a batch that the handle cuts
    into chunks (the last one partial)."""
    import ctypes as C
    import torch
    w, h, nf, b = 320, 240, 500, 5
    imgs = np.stack([synth.synth_image(w, h, 700 + i) for i in range(b)])
    ext = orb.ORBextractor(nf, 1.2, 8, 1, 20, max_width=w, max_height=h, max_batch=2)
    pageable = ext.extract_batch(imgs)
    pinned = torch.from_numpy(imgs).pin_memory()
    view = np.ctypeslib.as_array((C.c_uint8 * imgs.size).from_address(pinned.data_ptr())).reshape(imgs.shape)
    assert view.ctypes.data == pinned.data_ptr()
    locked = ext.extract_batch(view)
    assert len(pageable) == len(locked) == b
    for (kp0, d0), (kp1, d1) in zip(pageable, locked):
        assert len(kp0) > 0 and kp0.tobytes() == kp1.tobytes() and np.array_equal(d0, d1)


def test_large_host_batch_runs_in_pieces(orb, synth):
    """A host-buffer batch of 512 frames and more is cut into pieces whose copies run on a stream of their own under
    the kernels of the piece before: same keypoints and descriptors as the same frames through a handle that takes
    them in chunks of 100 (never in pieces), from pageable and from page-locked memory, with a frame count that is not
    a multiple of anything."""
    import ctypes as C
    import torch
    w, h, nf, b = 160, 120, 200, 603
    base = [synth.synth_image(w, h, 900 + i) for i in range(24)]
    imgs = np.stack([np.roll(base[i % 24], i // 24, axis=1) for i in range(b)])
    whole = orb.ORBextractor(nf, 1.2, 4, 1, 20, max_width=w, max_height=h, max_batch=b)
    chunked = orb.ORBextractor(nf, 1.2, 4, 1, 20, max_width=w, max_height=h, max_batch=100)
    ref = chunked.extract_batch(imgs)
    got = whole.extract_batch(imgs)
    pinned = torch.from_numpy(imgs).pin_memory()
    view = np.ctypeslib.as_array((C.c_uint8 * imgs.size).from_address(pinned.data_ptr())).reshape(imgs.shape)
    got_locked = whole.extract_batch(view)
    assert len(ref) == len(got) == len(got_locked) == b
    for (kp0, d0), (kp1, d1), (kp2, d2) in zip(ref, got, got_locked):
        assert len(kp0) > 0 and kp0.tobytes() == kp1.tobytes() == kp2.tobytes()
        assert np.array_equal(d0, d1) and np.array_equal(d0, d2)


def test_blur_on_its_own_stream_beside_the_selection(orb, orbref, synth, monkeypatch):
    """Batches of 64 frames and more run the Gaussian blur on a stream of its own beside the selection kernels (both
    follow the FAST kernel; the descriptor kernel waits for both).  Same keypoints and descriptors as with the blur in
    line (ORBHIP_BLUR_STREAM=0 when the handle is created), several calls in a row on one handle (the blurred arena is
    reused from call to call), and equal to the oracle on a sample of the frames."""
    w, h, nf, b = 320, 240, 500, 80
    base = [synth.synth_image(w, h, 1500 + i) for i in range(16)]
    batches = [np.stack([np.roll(base[(i + 3 * k) % 16], (i // 16) + k, axis=1) for i in range(b)]) for k in range(3)]
    beside = orb.ORBextractor(nf, 1.2, 8, 1, 20, max_width=w, max_height=h, max_batch=b)
    monkeypatch.setenv("ORBHIP_BLUR_STREAM", "0")
    inline = orb.ORBextractor(nf, 1.2, 8, 1, 20, max_width=w, max_height=h, max_batch=b)
    monkeypatch.delenv("ORBHIP_BLUR_STREAM")
    for imgs in batches:
        got, ref = beside.extract_batch(imgs), inline.extract_batch(imgs)
        assert len(got) == len(ref) == b
        for (kp0, d0), (kp1, d1) in zip(got, ref):
            assert len(kp0) > 0 and kp0.tobytes() == kp1.tobytes() and np.array_equal(d0, d1)
    ext = orbref.RefExtractor(nf, 1.2, 8, 1, 20)
    for i in (0, 37, 79):
        rkp, rdesc = ext(batches[-1][i])
        _assert_same(rkp, rdesc, *got[i])
