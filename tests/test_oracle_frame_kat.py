"""Known-answer tests of the oracle's frame grid, GetFeaturesInArea and windowed matchers
(oracle/frame_ref.cpp).  The reference holds no tests or fixtures for this path (parity
unpinned), so the expected values below are derived by hand from src/Frame.cc:201-278 and
src/ORBmatcher.cc:420-527,609-724."""
import numpy as np

INT_MAX = 2147483647
B640 = [0, 0, 640, 480]  # mfGridElementWidthInv = 64/640 = 0.1, mfGridElementHeightInv = 48/480 = 0.1


def kps(orbref, pts, octave=0, angle=0.0):
    kp = np.zeros(len(pts), orbref.KP_DTYPE)
    for i, p in enumerate(pts):
        kp["x"][i], kp["y"][i] = p[0], p[1]
        kp["octave"][i] = p[2] if len(p) > 2 else octave
        kp["angle"][i] = p[3] if len(p) > 3 else angle
    return kp


def bits(k):
    d = np.zeros(32, np.uint8)
    for i in range(k):
        d[i // 8] |= np.uint8(1 << (i % 8))
    return d


def test_pos_in_grid_rounds_half_away_and_rejects_the_far_edge(orbref):
    # x*0.1: 1.49 -> 1, 1.5 -> 2 (std::round), 63.5 -> 64 = FRAME_GRID_COLS -> in no cell; y 47.5 -> 48 -> none
    kp = kps(orbref, [(14.9, 0), (15.0, 0), (635.0, 10), (634.9, 10), (20, 475.0), (20, 474.9), (-5.1, 0), (-4.9, 0)])
    start, items = orbref.frame_grid(kp, B640)
    cell = {}
    for c in range(64 * 48):
        for i in items[start[c]:start[c + 1]]:
            cell[int(i)] = (c // 48, c % 48)
    assert cell == {0: (1, 0), 1: (2, 0), 3: (63, 1), 5: (2, 47), 7: (0, 0)}
    assert start[-1] == 5


def test_grid_items_keep_push_order(orbref):
    kp = kps(orbref, [(100, 100)] * 5 + [(50, 50)] * 2)
    start, items = orbref.frame_grid(kp, B640)
    assert list(items) == [5, 6, 0, 1, 2, 3, 4]  # cell (5,5) precedes cell (10,10)


def test_features_in_area_order_levels_and_closed_window(orbref):
    # index order deliberately differs from grid order: result is grid-column major, row minor
    kp = kps(orbref, [(100, 130, 0), (130, 100, 1), (100, 100, 0), (135, 100, 0), (135.5, 100, 0)])
    assert list(orbref.features_in_area(kp, B640, 115.0, 115.0, 20.0)) == [2, 0, 1, 3]  # |dx| = 20 is inside, 20.5 is not
    assert list(orbref.features_in_area(kp, B640, 115.0, 115.0, 20.0, 0, 0)) == [2, 0, 3]
    assert list(orbref.features_in_area(kp, B640, 115.0, 115.0, 20.0, 1, 3)) == [1]
    assert list(orbref.features_in_area(kp, B640, 115.0, 115.0, 14.0)) == []
    assert list(orbref.features_in_area(kp, B640, 2000.0, 115.0, 20.0)) == []   # nMinCellX >= FRAME_GRID_COLS
    assert list(orbref.features_in_area(kp, B640, -500.0, 115.0, 20.0)) == []   # nMaxCellX < 0


def test_search_for_initialization_later_keypoint_steals_or_is_blocked(orbref):
    kp2 = kps(orbref, [(100, 100)])
    d2 = np.stack([bits(0)])
    kp1 = kps(orbref, [(101, 100), (99, 100)])
    prev = np.array([[101, 100], [99, 100]], np.float32)
    # distances 10 then 5: the second takes the feature, the first loses it
    n, m12, pv = orbref.search_for_initialization(kp1, np.stack([bits(10), bits(5)]), kp2, d2, B640, prev, 100, 0.9, False)
    assert n == 1 and list(m12) == [-1, 0]
    assert pv[1, 0] == 100 and pv[0, 0] == 101
    # distances 10 then 12: vMatchedDistance[0] = 10 <= 12 blocks the candidate
    n, m12, _ = orbref.search_for_initialization(kp1, np.stack([bits(10), bits(12)]), kp2, d2, B640, prev, 100, 0.9, False)
    assert n == 1 and list(m12) == [0, -1]
    # TH_LOW = 20 is inclusive; 21 is rejected
    n, m12, _ = orbref.search_for_initialization(kp1[:1], np.stack([bits(20)]), kp2, d2, B640, prev[:1], 100, 0.9, False)
    assert n == 1
    n, m12, _ = orbref.search_for_initialization(kp1[:1], np.stack([bits(21)]), kp2, d2, B640, prev[:1], 100, 0.9, False)
    assert n == 0
    # only level-0 keypoints of F1 are matched
    n, _, _ = orbref.search_for_initialization(kps(orbref, [(101, 100, 1)]), np.stack([bits(3)]), kp2, d2, B640, prev[:1], 100, 0.9, False)
    assert n == 0


def test_search_for_initialization_ratio_is_strict(orbref):
    kp2 = kps(orbref, [(100, 100), (102, 100)])
    kp1 = kps(orbref, [(101, 100)])
    prev = np.array([[101, 100]], np.float32)
    # best 9, second 10: 9 < 10 * 0.9 is false
    n, _, _ = orbref.search_for_initialization(kp1, np.stack([bits(0)]), kp2, np.stack([bits(9), bits(10)]), B640, prev, 100, 0.9, False)
    assert n == 0
    n, m12, _ = orbref.search_for_initialization(kp1, np.stack([bits(0)]), kp2, np.stack([bits(8), bits(10)]), B640, prev, 100, 0.9, False)
    assert n == 1 and m12[0] == 0


def test_window_search_skips_taken_features_and_accepts_equal_ratio(orbref):
    kp2 = kps(orbref, [(100, 100), (104, 100)])
    d2 = np.stack([bits(0), bits(6)])
    kp1 = kps(orbref, [(101, 100), (102, 100)])
    d1 = np.stack([bits(0), bits(0)])
    v = np.ones(2, np.uint8)
    # first keypoint: 0 <= 6*0.6 takes feature 0; second: feature 0 taken, feature 1 alone (d2 = INT_MAX) accepted
    n, m21 = orbref.window_search(kp1, d1, v, kp2, d2, B640, 50, -1, INT_MAX, 0.6, False)
    assert n == 2 and list(m21) == [0, 1]
    # a keypoint without a good map point is not matched
    n, m21 = orbref.window_search(kp1, d1, np.array([0, 1], np.uint8), kp2, d2, B640, 50, -1, INT_MAX, 0.6, False)
    assert n == 1 and list(m21) == [1, -1]
    # "bestDist <= bestDist2 * ratio" is not strict: 6 <= 10 * 0.6
    n, m21 = orbref.window_search(kp1[:1], np.stack([bits(0)]), v[:1], kp2, np.stack([bits(6), bits(10)]), B640, 50, -1, INT_MAX, 0.6, False)
    assert n == 1 and m21[0] == 0
    # minScaleLevel = 1 filters level-0 keypoints of F1; minScaleLevel = 0 does not filter
    n, _ = orbref.window_search(kp1, d1, v, kp2, d2, B640, 50, 1, INT_MAX, 0.6, False)
    assert n == 0


def test_rotation_histogram_drops_the_outlier_bin(orbref):
    # 12 matches with rotation 3 deg (bin round(0.1) = 0), one with 100 deg (bin round(3.33) = 3):
    # max2 = 1 < 0.1 * 12 -> only bin 0 survives
    n = 13
    pts = [(40 * (i % 8) + 60, 60 * (i // 8) + 60) for i in range(n)]
    kp1 = kps(orbref, [(x, y, 0, 103.0 if i < 12 else 200.0) for i, (x, y) in enumerate(pts)])
    kp2 = kps(orbref, [(x, y, 0, 100.0) for (x, y) in pts])
    rng = np.random.default_rng(1)
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    prev = np.array(pts, np.float32)
    nm, m12, _ = orbref.search_for_initialization(kp1, d, kp2, d, B640, prev, 10, 0.9, True)
    assert nm == 12 and list(m12) == list(range(12)) + [-1]
    nm, m12, _ = orbref.search_for_initialization(kp1, d, kp2, d, B640, prev, 10, 0.9, False)
    assert nm == 13
    nm, m21 = orbref.window_search(kp1, d, np.ones(n, np.uint8), kp2, d, B640, 10, -1, INT_MAX, 0.6, True)
    assert nm == 12 and m21[12] == -1
    # negative difference wraps: 10 - 100 + 360 = 270 -> bin 9
    assert orbref.lib().orbref_rot_bin(10.0, 100.0) == 9


def test_distinctive_descriptor_median_rank_and_first_minimum(orbref):
    # N = 4: rank (size_t)(0.5 * 3) = 1 of the sorted row (which contains the 0 of the diagonal)
    # rows: a = 0 bits, b = 2 bits, c = 4 bits (supersets), d = 40 bits
    d = np.stack([bits(0), bits(2), bits(4), bits(40)])
    # row a: [0,2,4,40] -> 2; row b: [0,2,2,38] -> 2; row c: [0,2,4,36] -> 2; row d: [0,36,38,40] -> 36
    assert orbref.distinctive_descriptor(d) == (0, 2)  # first of the tied rows
    assert orbref.distinctive_descriptor(d[::-1]) == (1, 2)
    assert orbref.distinctive_descriptor(d[:1]) == (0, 0)
    assert orbref.distinctive_descriptor(d[:2]) == (0, 0)  # N = 2: rank 0 is the diagonal
    assert orbref.distinctive_descriptor(d[:0]) == (-1, INT_MAX)


def test_bow_transform_by_hand(orbref):
    """Depth-2 binary tree: root 0 -> {1, 4}; 1 -> {2, 3}; 4 -> {5, 6}.  Words (leaves in id order):
    2 -> 0, 3 -> 1, 5 -> 2, 6 -> 3."""
    parent = np.array([0, 0, 1, 1, 0, 4, 4], np.int32)
    desc = np.stack([bits(0), bits(0), bits(0), bits(8), bits(200), bits(200), bits(200)])
    weight = np.array([0, 0, 1.5, 2.0, 0, 0.0, 4.0])
    voc = orbref.RefVocabulary(parent, desc, weight, 2)
    # bits(3): nearer to node 1 (3) than node 4 (197); then node 2 (3) vs node 3 (5) -> word 0
    # bits(5): node 2 (5) vs node 3 (3) -> word 1; bits(4): tie 4 / 4 -> first child, word 0
    # bits(150): node 4; children 5 and 6 identical -> first (word 2, stopped: weight 0)
    f = np.stack([bits(3), bits(5), bits(4), bits(150)])
    word, node, w = voc.transform_features(f, 1)  # node at level L - 1 = 1
    assert list(word) == [0, 1, 0, 2] and list(node) == [1, 1, 1, 4] and list(w) == [1.5, 2.0, 1.5, 0.0]
    assert list(voc.transform_features(f, 2)[1]) == [0, 0, 0, 0]   # level 0 = root
    assert list(voc.transform_features(f, 0)[1]) == [2, 3, 2, 5]   # level L = the leaves
    (bw, bv), (fn, fs, ff) = voc.transform(f, 1)
    # TF_IDF + L1: word 0 = 1.5 + 1.5, word 1 = 2.0, stopped word absent; normalised by 5.0
    assert list(bw) == [0, 1] and list(bv) == [3.0 / 5.0, 2.0 / 5.0]
    assert list(fn) == [1] and list(ff) == [0, 1, 2]
    binary = orbref.RefVocabulary(parent, desc, weight, 2, weighting=3, scoring=5)
    (bw, bv), _ = binary.transform(f, 1)
    assert list(bw) == [0, 1] and list(bv) == [1.5, 2.0]           # addIfNotExist, no normalisation


def test_search_by_bow_by_hand(orbref):
    """Two nodes.  Node 5: keyframe features [0, 1] vs frame features [0, 1]; node 9: [2] vs [2].
    Feature 0 of the keyframe takes frame feature 1 (distance 2 vs 12); feature 1 then sees only frame
    feature 0 (a single candidate: second best INT_MAX) at distance 20 -> accepted by the KeyFrame/Frame
    form (<= TH_LOW), rejected by the KeyFrame/KeyFrame form (< TH_LOW)."""
    kp = kps(orbref, [(10, 10, 0, 0.0)] * 3)
    d1 = np.stack([bits(0), bits(0), bits(7)])
    d2 = np.stack([bits(12) if False else bits(20), bits(2), bits(9)])
    # distances: kf0-f0 = 20, kf0-f1 = 2; kf1-f0 = 20 (f1 taken); kf2-f2 = 2
    fv = (np.array([5, 9], np.int32), np.array([0, 2, 3], np.int32), np.array([0, 1, 2], np.int32))
    ones = np.ones(3, np.uint8)
    n, mF = orbref.search_by_bow_kf_frame(kp, d1, ones, fv, kp, d2, fv, 0.7, False)
    assert n == 3 and list(mF) == [1, 0, 2]
    n, m12 = orbref.search_by_bow_kf_kf(kp, d1, ones, fv, kp, d2, ones, fv, 0.7, False)
    assert n == 2 and list(m12) == [1, -1, 2]
    # a keyframe feature without a good map point is skipped; a bad map point in keyframe 2 is skipped
    n, m12 = orbref.search_by_bow_kf_kf(kp, d1, np.array([0, 1, 1], np.uint8), fv, kp, d2, np.array([1, 0, 1], np.uint8), fv,
                                        0.7, False)
    assert n == 1 and list(m12) == [-1, -1, 2]
    # nodes present in only one FeatureVector are skipped by the lower_bound stepping
    fvb = (np.array([5, 7, 9], np.int32), np.array([0, 1, 2, 3], np.int32), np.array([1, 0, 2], np.int32))
    n, m12 = orbref.search_by_bow_kf_kf(kp, d1, ones, fv, kp, d2, ones, fvb, 0.7, False)
    assert n == 2 and list(m12) == [1, -1, 2]  # node 5 of keyframe 2 now holds only feature 1


def test_projection_last_frame_by_hand(orbref):
    """SearchByProjection(CurrentFrame, LastFrame, th) (src/ORBmatcher.cc:1518-1631), camera fx = fy = 100,
    cx = cy = 50, Tcw = [I | (0, 0, 1)], th = 5.
    point 0 (0.25, 0.5, 1) -> camera (0.25, 0.5, 2) -> u = 100*0.25*0.5 + 50 = 62.5, v = 75; octave 1:
            radius = 5 * 1.2f = 6.0 (the product is an exact tie, rounded to even), levels 0..2
    point 1 skipped (no map point / outlier);  point 2 (-1, 0, 1) -> u = 0: "u < mnMinX" is false, kept; octave 0
    point 3 (13, 0, 1) -> u = 700 > mnMaxX: dropped."""
    sf = np.array([1.0, 1.2, 1.44], np.float32)
    T = np.eye(4, dtype=np.float32)
    T[2, 3] = 1.0
    last = kps(orbref, [(0, 0, 1, 10.0), (0, 0, 0, 0.0), (0, 0, 0, 200.0), (0, 0, 0, 0.0)])
    pos = np.array([[0.25, 0.5, 1], [0, 0, 1], [-1, 0, 1], [13, 0, 1]], np.float32)
    dlast = np.stack([bits(0), bits(0), bits(10), bits(0)])
    #             a: in window   b: level 3     c: taken        d: 6.5 px off   e: near point 2
    cur = kps(orbref, [(60, 75, 1), (64, 79, 3), (68, 75, 1), (69, 75, 1), (2, 52, 1)])
    dcur = np.stack([bits(3), bits(0), bits(1), bits(0), bits(110)])
    taken = np.array([0, 0, 1, 0, 0], np.uint8)
    for ori in (False, True):
        n, match, probes = orbref.sbp_last_frame(cur, dcur, B640, taken, (100.0, 100.0, 50.0, 50.0), T, sf, last, dlast,
                                                 [0, 1, 0, 0], pos, 5.0, ori)
        assert list(probes["valid"]) == [1, 0, 1, 0]
        assert (probes["u"][0], probes["v"][0], probes["r"][0]) == (62.5, 75.0, 6.0)
        assert (probes["min_level"][0], probes["max_level"][0], probes["angle"][0]) == (0, 2, 10.0)
        assert (probes["u"][2], probes["v"][2], probes["r"][2], probes["min_level"][2], probes["max_level"][2]) == (0.0, 50.0, 5.0, -1, 1)
        # point 0: a (3 bits) wins, c is taken, b and d are outside; point 2: e at exactly TH_HIGH = 100 is accepted
        assert n == 2 and list(match) == [0, -1, -1, -1, 2]
    # one bit further is past TH_HIGH
    dcur[4] = bits(111)
    n, match, _ = orbref.sbp_last_frame(cur, dcur, B640, taken, (100.0, 100.0, 50.0, 50.0), T, sf, last, dlast, [0, 1, 0, 0],
                                        pos, 5.0, False)
    assert n == 1 and list(match) == [0, -1, -1, -1, -1]


def test_projection_local_map_by_hand(orbref):
    """SearchByProjection(Frame&, vector<MapPoint*>&, th) (src/ORBmatcher.cc:52-128), th = 1, ratio 0.8.
    points 0, 1: both project to (100, 100), level 2 (levels 1..2, radius 2.5 * 1.44): point 0 takes a (10 bits; the
            second best c is on another level, so no ratio test); point 1 then sees b (12, level 2) and c (11, level 1):
            takes c, again across levels
    point 2: d, e on one level at 10 and 11 bits: 10 > 0.8 * 11 -> rejected
    point 3: f, g on one level at 8 and 11 bits: 8 <= 8.8 -> accepted
    point 4: mTrackViewCos = 0.998f, which is greater than the double 0.998: radius 2.5, h at 2.6 px stays outside
    point 5: not in view; point 6: bad."""
    sf = np.array([1.0, 1.2, 1.44, 1.728], np.float32)
    cur = kps(orbref, [(100, 100, 2), (101, 100, 2), (102, 100, 1), (200, 200, 3), (201, 200, 3), (300, 300, 3), (301, 300, 3),
                       (402.6, 400, 0)])
    dcur = np.stack([bits(10), bits(12), bits(11), bits(10), bits(11), bits(8), bits(11), bits(0)])
    px = np.array([100, 100, 200, 300, 400, 100, 100], np.float32)
    level = np.array([2, 2, 3, 3, 0, 2, 2], np.int32)
    cosv = np.array([0.999, 0.999, 0.999, 0.999, 0.998, 0.999, 0.999], np.float32)
    in_view = np.array([1, 1, 1, 1, 1, 0, 1], np.uint8)
    bad = np.array([0, 0, 0, 0, 0, 0, 1], np.uint8)
    mp_desc = np.stack([bits(0)] * 7)
    n, match, probes = orbref.sbp_local_map(cur, dcur, B640, sf, None, in_view, bad, level, cosv, px, px, mp_desc, 1.0, 0.8)
    assert list(probes["valid"]) == [1, 1, 1, 1, 1, 0, 0]
    assert probes["r"][0] == np.float32(2.5) * sf[2] and probes["r"][4] == 2.5
    assert (probes["min_level"][4], probes["max_level"][4]) == (-1, 0)
    assert n == 3 and list(match) == [0, -1, 1, -1, -1, 3, -1, -1]
    # th = 3 multiplies the radius: point 4 now reaches h
    n, match, probes = orbref.sbp_local_map(cur, dcur, B640, sf, None, in_view, bad, level, cosv, px, px, mp_desc, 3.0, 0.8)
    assert probes["r"][4] == 7.5 and match[7] == 4


def test_projection_sim3_and_keyframe_by_hand(orbref):
    """Scw = 2 * [I | (0, 0, 0.5)]: scw = 2, Rcw = I, tcw = (0, 0, 0.5), Ow = (0, 0, -0.5).
    point (0.5, 0, 1.5) -> camera (0.5, 0, 2) -> u = 100*0.25 + 50 = 75, v = 50; PO = (0.5, 0, 2), dist = sqrt(4.25);
    minDistance = dist / 1.3 -> ratio 1.3 -> lower_bound over (1, 1.2, 1.44) = level 2; radius = 10 * 1.44;
    the keypoint test is a plain level range 1..2 after an unfiltered KeyFrame::GetFeaturesInArea."""
    sf = np.array([1.0, 1.2, 1.44], np.float32)
    S = np.eye(4, dtype=np.float32) * 2
    S[3, 3] = 1
    S[2, 3] = 1.0
    pos = np.array([[0.5, 0, 1.5]], np.float32)
    dist = np.float32(np.sqrt(4.25))
    min_d = np.array([dist / np.float32(1.3)], np.float32)
    max_d = min_d * 10
    normal = np.array([[0, 0, 1]], np.float32)  # PO.n = 2 >= 0.5 * dist
    cur = kps(orbref, [(75, 50, 0), (80, 55, 1), (76, 50, 2)])
    dcur = np.stack([bits(0), bits(20), bits(21)])
    K = (100.0, 100.0, 50.0, 50.0)
    n, match, probes = orbref.sbp_sim3(cur, dcur, B640, K, sf, S, None, [0], pos, normal, min_d, max_d, bits(0)[None], 10)
    assert (probes["u"][0], probes["v"][0], probes["min_level"][0], probes["max_level"][0]) == (75.0, 50.0, 1, 2)
    assert probes["r"][0] == np.float32(10) * sf[2]
    assert n == 1 and list(match) == [-1, 0, -1]  # level 0 is out of range; 20 bits = TH_LOW accepted, 21 would not be
    n, match, _ = orbref.sbp_sim3(cur, dcur, B640, K, sf, S, [0, 1, 0], [0], pos, normal, min_d, max_d, bits(0)[None], 10)
    assert n == 0 and list(match) == [-1, -1, -1]  # the 20-bit feature is already matched; the 21-bit one is too far
    # viewing angle: PO.n < 0.5 * dist drops the point; so does a depth outside [minDistance, maxDistance]
    n, _, probes = orbref.sbp_sim3(cur, dcur, B640, K, sf, S, None, [0], pos, np.array([[1, 0, 0]], np.float32), min_d, max_d,
                                   bits(0)[None], 10)
    assert n == 0 and probes["valid"][0] == 0
    n, _, probes = orbref.sbp_sim3(cur, dcur, B640, K, sf, S, None, [0], pos, normal, min_d * 3, max_d * 3, bits(0)[None], 10)
    assert n == 0 and probes["valid"][0] == 0
    # the relocalisation overload on the same geometry (Tcw = [I | (0, 0, 0.5)]): levels 1..3, threshold ORBdist
    T = np.eye(4, dtype=np.float32)
    T[2, 3] = 0.5
    n, match, probes = orbref.sbp_keyframe(cur, dcur, B640, None, K, T, sf, [0], pos, min_d, bits(0)[None], [30.0], 10.0, 20, False)
    assert (probes["u"][0], probes["v"][0], probes["min_level"][0], probes["max_level"][0], probes["angle"][0]) == (75.0, 50.0, 1, 3, 30.0)
    assert n == 1 and list(match) == [-1, 0, -1]
    n, match, _ = orbref.sbp_keyframe(cur, dcur, B640, None, K, T, sf, [0], pos, min_d, bits(0)[None], [30.0], 10.0, 19, False)
    assert n == 0


def test_fuse_by_hand(orbref):
    """The geometry of the Sim3 projection test above (camera at (0, 0, -0.5), point (0.5, 0, 1.5) -> (75, 50), levels
    1..2, radius th * 1.44).  Fuse has no "already matched" skip: two equal map points find the same feature, and the
    first of two equally distant features (GetFeaturesInArea order) wins."""
    sf = np.array([1.0, 1.2, 1.44], np.float32)
    T = np.eye(4, dtype=np.float32)
    T[2, 3] = 0.5
    S = np.eye(4, dtype=np.float32) * 2
    S[3, 3] = 1
    S[2, 3] = 1.0
    pos = np.array([[0.5, 0, 1.5], [0.5, 0, 1.5]], np.float32)
    dist = np.float32(np.sqrt(4.25))
    min_d = np.array([dist / np.float32(1.3)] * 2, np.float32)
    max_d = min_d * 10
    normal = np.array([[0, 0, 1], [0, 0, 1]], np.float32)
    K = (100.0, 100.0, 50.0, 50.0)
    kf = kps(orbref, [(75, 50, 0), (80, 55, 1), (76, 50, 2), (70, 45, 2)])
    dkf = np.stack([bits(0), bits(20), bits(21), bits(20)])
    mp = np.stack([bits(0), bits(0)])
    for T_, sim3 in ((T, False), (S, True)):
        n, bi, bd, probes = orbref.fuse(kf, dkf, B640, K, sf, T_, [0, 0], pos, normal, min_d, max_d, mp, 2.5, sim3=sim3)
        assert (probes["u"][0], probes["v"][0], probes["min_level"][0], probes["max_level"][0]) == (75.0, 50.0, 1, 2)
        assert probes["r"][0] == np.float32(2.5) * sf[2]
        # radius 3.6 around (75, 50): features 0 (level 0, filtered) and 2 (1 px away, 21 bits) are inside
        assert n == 0 and list(bi) == [2, 2] and list(bd) == [21, 21]
        n, bi, bd, _ = orbref.fuse(kf, dkf, B640, K, sf, T_, [0, 0], pos, normal, min_d, max_d, mp, 5.0, sim3=sim3)
        # radius 7.2: features 1 (20 bits, cell column of x = 80) and 3 (20 bits, column of x = 70) tie; the area is
        # walked by grid column, so feature 3 comes first
        assert n == 2 and list(bi) == [3, 3] and list(bd) == [20, 20]
        n, bi, bd, probes = orbref.fuse(kf, dkf, B640, K, sf, T_, [1, 0], pos, np.array([[0, 0, 1], [1, 0, 0]], np.float32), min_d,
                                        max_d, mp, 5.0, sim3=sim3)
        assert n == 0 and list(bi) == [-1, -1] and list(bd) == [INT_MAX, INT_MAX] and list(probes["valid"]) == [0, 0]
    # a keyframe whose only feature in the radius fails the level test: the scan runs and finds nothing
    n, bi, bd, probes = orbref.fuse(kf[:1], dkf[:1], B640, K, sf, T, [0], pos[:1], normal[:1], min_d[:1], max_d[:1], mp[:1], 2.5)
    assert n == 0 and probes["valid"][0] == 1 and bi[0] == -1 and bd[0] == INT_MAX


def test_search_by_sim3_by_hand(orbref):
    """Both cameras at the origin looking down z, identity Sim3 (s12 = 1, R12 = I, t12 = 0): a map point of one
    keyframe projects to the same pixel in the other.  Point A (0.5, 0, 2) -> (75, 50); point B (1, 0, 2) -> (100, 50).
    Keyframe 1 holds A at feature 0 and B at feature 1; keyframe 2 holds A at feature 1 and nothing else.  A agrees in
    both directions; B finds keyframe 2's feature 0 (30 bits <= TH_HIGH) but that feature has no map point to answer."""
    sf = np.array([1.0, 1.2, 1.44], np.float32)
    I4 = np.eye(4, dtype=np.float32)
    K = (100.0, 100.0, 50.0, 50.0)
    A, B = [0.5, 0, 2], [1, 0, 2]
    dA = np.float32(np.sqrt(4.25))
    dB = np.float32(np.sqrt(5.0))
    kf1 = kps(orbref, [(75, 50, 1), (100, 50, 1)])
    kf2 = kps(orbref, [(101, 50, 1), (76, 51, 1)])
    d1 = np.stack([bits(3), bits(0)])
    d2 = np.stack([bits(30), bits(5)])
    pos1 = np.array([A, B], np.float32)
    pos2 = np.array([[0, 0, 0], A], np.float32)
    min1 = np.array([dA / np.float32(1.1), dB / np.float32(1.1)], np.float32)  # ratio 1.1 -> level 1, levels 0..1
    min2 = np.array([1, dA / np.float32(1.1)], np.float32)
    mp1 = np.stack([bits(4), bits(0)])
    mp2 = np.stack([bits(0), bits(4)])
    args = (kf1, d1, kf2, d2, B640, K, sf, I4, I4, 1.0, np.eye(3, dtype=np.float32), np.zeros(3, np.float32), 7.5)
    n, m12, vn1, vn2, pr1, pr2 = orbref.search_by_sim3(*args, [0, 0], pos1, min1, min1 * 10, mp1, [1, 0], pos2, min2, min2 * 10, mp2)
    assert (pr1["u"][0], pr1["v"][0], pr1["u"][1], pr2["u"][1]) == (75.0, 50.0, 100.0, 75.0)
    assert pr1["r"][0] == np.float32(7.5) * sf[1] and (pr1["min_level"][0], pr1["max_level"][0]) == (0, 1)
    assert list(vn1) == [1, 0] and list(vn2) == [-1, 0]  # A -> feature 1 of keyframe 2 (1 bit), B -> feature 0 (30 bits)
    assert n == 1 and list(m12) == [1, -1]
    # beyond TH_HIGH nothing is recorded: B's descriptor against 30 + 101 bits
    d2b = d2.copy()
    d2b[0] = bits(131)
    n, m12, vn1, _, _, _ = orbref.search_by_sim3(kf1, d1, kf2, d2b, *args[4:], [0, 0], pos1, min1, min1 * 10, mp1, [1, 0], pos2, min2,
                                                 min2 * 10, mp2)
    assert list(vn1) == [1, -1] and n == 1
    # a scale: s12 = 2 maps keyframe 2's point (0.25, 0, 1) to (0.5, 0, 2) in camera 1 and camera 1's A to (0.25, 0, 1):
    # the same pixels, but the distances used for the level prediction halve in keyframe 2
    pos2s = np.array([[0, 0, 0], [0.25, 0, 1]], np.float32)
    n, m12, vn1, vn2, pr1, pr2 = orbref.search_by_sim3(kf1, d1, kf2, d2, B640, K, sf, I4, I4, 2.0, np.eye(3, dtype=np.float32),
                                                       np.zeros(3, np.float32), 7.5, [0, 1], pos1, min1 / 2, min1 * 10, mp1, [1, 0],
                                                       pos2s, min2, min2 * 10, mp2)
    assert (pr1["u"][0], pr1["v"][0], pr2["u"][1], pr2["v"][1]) == (75.0, 50.0, 75.0, 50.0)
    assert n == 1 and list(m12) == [1, -1]


def test_search_for_triangulation_by_hand(orbref):
    """F12 = [[0,0,0],[0,0,-1],[0,1,0]]: a = 0, b = 1, c = -y1, so the epipolar line of (x1, y1) is y2 = y1 and
    dsqr = (y2 - y1)^2, accepted below 3.84 * sigma2[octave of keypoint 2] (src/ORBmatcher.cc:139-156).  All features
    under node 7; node 3 exists only in keyframe 1 and node 9 only in keyframe 2 (the lower_bound skips of :976-983)."""
    F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    sigma2 = np.array([1.0, 1.44], np.float32)
    kp1 = kps(orbref, [(10, 100), (20, 100), (30, 100), (40, 100)])
    d1 = np.stack([bits(0), bits(0), bits(0), bits(0)])
    kp2 = kps(orbref, [(11, 103, 0), (12, 101.9, 0), (13, 102.2, 1), (14, 100, 0), (15, 100, 0)])
    d2 = np.stack([bits(2), bits(4), bits(4), bits(5), bits(0)])
    fv1 = ([3, 7], [0, 1, 4], [3, 0, 1, 2])
    fv2 = ([7, 9], [0, 4, 5], [0, 1, 2, 3, 4])
    free1, free2 = [1, 1, 0, 1], [1, 1, 1, 1, 1]
    # feature 0: candidates (2, #0) off the line (dy = 3), (4, #1) on it (1.9^2 = 3.61 < 3.84) -> 0 -> 1.
    # feature 1: #1 is taken; (2, #0) off, (4, #2) on the line of level 1 (2.2^2 = 4.84 < 5.5296) -> 1 -> 2.
    # feature 2 holds a map point; feature 3 sits under node 3, which keyframe 2 does not have, although #4 (node 9) is identical.
    n, m12 = orbref.search_for_triangulation(kp1, d1, free1, fv1, kp2, d2, free2, fv2, F, sigma2, False)
    assert n == 2 and list(m12) == [1, 2, -1, -1]
    # with sigma2 = 1 on level 1, #2 is off the line as well; (5, #3) lies beyond DistTh = 2 * 2 and is never tried
    n, m12 = orbref.search_for_triangulation(kp1, d1, free1, fv1, kp2, d2, free2, fv2, F, np.array([1.0, 1.0], np.float32), False)
    assert n == 1 and list(m12) == [1, -1, -1, -1]
    # #0 holds a map point: BestDist = 4, DistTh = 8, and for feature 1 (5, #3) is now inside the bound and on the line
    n, m12 = orbref.search_for_triangulation(kp1, d1, free1, fv1, kp2, d2, [0, 1, 1, 1, 1], fv2, F, np.array([1.0, 1.0], np.float32),
                                             False)
    assert n == 2 and list(m12) == [1, 3, -1, -1]
    # distances above TH_LOW = 20 are no candidates at all; 20 itself is
    d2b = np.stack([bits(21), bits(21), bits(21), bits(20), bits(0)])
    n, m12 = orbref.search_for_triangulation(kp1, d1, free1, fv1, kp2, d2b, free2, fv2, F, sigma2, False)
    assert n == 1 and list(m12) == [3, -1, -1, -1]
    # den == 0 (a zero matrix) rejects everything
    n, m12 = orbref.search_for_triangulation(kp1, d1, free1, fv1, kp2, d2, free2, fv2, np.zeros((3, 3), np.float32), sigma2, False)
    assert n == 0 and list(m12) == [-1, -1, -1, -1]


def test_is_in_frustum_by_hand(orbref):
    """Tcw = [I | (0, 0, 0.5)] (camera centre (0, 0, -0.5)), K = (100, 100, 50, 50).  Point (0.5, 0, 1.5) -> camera
    (0.5, 0, 2) -> (75, 50); PO = (0.5, 0, 2), dist = sqrt(4.25); normal (0, 0, 1) -> viewCos = 2 / sqrt(4.25) = 0.9701;
    minDistance = dist / 1.3 -> level 2 of (1, 1.2, 1.44)."""
    sf = np.array([1.0, 1.2, 1.44], np.float32)
    T = np.eye(4, dtype=np.float32)
    T[2, 3] = 0.5
    K = (100.0, 100.0, 50.0, 50.0)
    dist = np.float32(np.sqrt(4.25))
    pos = np.array([[0.5, 0, 1.5],      # in view
                    [0.5, 0, -1.5],     # behind the camera
                    [11.8, 0, 1.5],     # u = 640 = mnMaxX: the bounds test is closed (u > mnMaxX rejects)
                    [11.9, 0, 1.5],     # u = 645: outside
                    [0.5, 0, 1.5],      # too close: dist < minDistance
                    [0.5, 0, 1.5]],     # far beyond the last level: clamped to level 2
                   np.float32)
    normal = np.tile(np.array([0, 0, 1], np.float32), (6, 1))
    normal[2:4] = np.array([11.8, 0, 2], np.float32) / np.float32(np.sqrt(11.8 ** 2 + 4))  # seen head-on
    min_d = np.array([dist / np.float32(1.3)] * 4 + [dist * 2, dist / 5], np.float32)
    max_d = min_d * 100
    iv, px, py, lv, vc = orbref.is_in_frustum(B640, K, T, sf, pos, normal, min_d, max_d, 0.5)
    assert list(iv) == [1, 0, 1, 0, 0, 1]
    assert (px[0], py[0], lv[0]) == (75.0, 50.0, 2) and px[2] == 640.0 and lv[5] == 2
    assert vc[0] == np.float32(2.0 / np.float64(dist))  # PO.dot(Pn) in double over the distance already rounded to float
    # the viewing-angle limit
    iv, _, _, _, _ = orbref.is_in_frustum(B640, K, T, sf, pos[:1], normal[:1], min_d[:1], max_d[:1], 0.98)
    assert list(iv) == [0]
    # RadiusByViewingCos: 0.9701 <= 0.998 -> 4.0; th = 1 leaves it, th = 3 multiplies (src/ORBmatcher.cc:63-81)
    kf = kps(orbref, [(75, 50, 2)])
    for th, want in ((1.0, np.float32(4.0) * sf[2]), (3.0, np.float32(4.0) * np.float32(3.0) * sf[2])):
        _, _, probes = orbref.sbp_local_map(kf, bits(0)[None], B640, sf, None, [1], [0], [2], vc[:1], px[:1], py[:1], bits(0)[None], th,
                                            0.8)
        assert probes["r"][0] == want and (probes["min_level"][0], probes["max_level"][0]) == (1, 2)


def test_projection_frame_to_frame_by_hand(orbref):
    """SearchByProjection(F1, F2, windowSize, ...) (src/ORBmatcher.cc:530-605): same-level window, ratio with <=."""
    T = np.eye(4, dtype=np.float32)
    K = (100.0, 100.0, 50.0, 50.0)
    kp1 = kps(orbref, [(0, 0, 1), (0, 0, 1), (0, 0, 2)])
    pos1 = np.array([[0.5, 0.5, 1], [0.5, 0.5, 1], [2, 2, 1]], np.float32)  # -> (100, 100), (100, 100), (250, 250)
    d1 = np.stack([bits(0), bits(0), bits(0)])
    kp2 = kps(orbref, [(101, 100, 1), (99, 100, 1), (250, 250, 1), (251, 250, 2)])
    d2 = np.stack([bits(8), bits(10), bits(0), bits(30)])
    # point 0: 8 vs 10 bits, 8 <= 10 * 0.8 -> takes feature 0; point 1: only feature 1 is left (second = INT_MAX) -> takes it;
    # point 2 (level 2): feature 2 is on level 1, feature 3 at 30 bits -> accepted
    n, match, probes = orbref.sbp_frames(kp1, d1, [0, 0, 0], pos1, T, K, kp2, d2, B640, None, 5, 0.8)
    assert (probes["u"][2], probes["v"][2], probes["r"][2], probes["min_level"][2], probes["max_level"][2]) == (250.0, 250.0, 5.0, 2, 2)
    assert n == 3 and list(match) == [0, 1, -1, 2]
    d2[1] = bits(9)  # 8 > 9 * 0.8: point 0 is rejected, point 1 sees the same pair and is rejected as well
    n, match, _ = orbref.sbp_frames(kp1, d1, [0, 0, 0], pos1, T, K, kp2, d2, B640, None, 5, 0.8)
    assert n == 1 and list(match) == [-1, -1, -1, 2]


def test_undistort_points_known_answers(orbref):
    """cv::undistortPoints as used by Frame::UndistortKeyPoints / ComputeImageBounds (src/Frame.cc:290-350).
    (1) the principal point has x = y = 0 and stays put; (2) k1 == 0 copies the keypoints whatever the other
    coefficients say (:292-296); (3) a point pushed through the forward Brown model comes back (five fixed-point
    iterations: 1e-3 px at mild distortion); (4) a pure radial k1 with y = 0: one iteration by hand."""
    K = (500.0, 500.0, 320.0, 240.0)
    kp = kps(orbref, [(320, 240), (100, 50), (600, 400)])
    kp["response"], kp["size"] = [3, 4, 5], 31
    un = orbref.undistort_keypoints(kp, K, (0.1, 0.01, 0.001, -0.002))
    assert (un["x"][0], un["y"][0]) == (320.0, 240.0)
    assert list(un["response"]) == [3, 4, 5] and list(un["octave"]) == list(kp["octave"])
    same = orbref.undistort_keypoints(kp, K, (0.0, 0.5, 0.1, 0.1))
    assert np.array_equal(same.view(np.uint32), kp.view(np.uint32))
    # forward model in double
    k1, k2, p1, p2 = -0.12, 0.03, 0.0007, -0.0004
    rng = np.random.default_rng(3)
    xn, yn = rng.uniform(-0.5, 0.5, 200), rng.uniform(-0.4, 0.4, 200)
    r2 = xn * xn + yn * yn
    cd = 1 + k1 * r2 + k2 * r2 * r2
    xd = xn * cd + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)
    yd = yn * cd + p1 * (r2 + 2 * yn * yn) + 2 * p2 * xn * yn
    kpd = kps(orbref, list(zip(500 * xd + 320, 500 * yd + 240)))
    un = orbref.undistort_keypoints(kpd, K, (k1, k2, p1, p2))
    assert np.abs(un["x"] - (500 * xn + 320)).max() < 2e-3 and np.abs(un["y"] - (500 * yn + 240)).max() < 2e-3
    # by hand: K = (1, 1, 0, 0), k1 = 0.5, point (0.5, 0): x <- x0 / (1 + k1 x^2), five times from x = x0
    x = 0.5
    for _ in range(5):
        x = 0.5 * (1.0 / (1.0 + 0.5 * (x * x)))
    un = orbref.undistort_keypoints(kps(orbref, [(0.5, 0.0)]), (1.0, 1.0, 0.0, 0.0), (0.5, 0.0, 0.0, 0.0))
    assert un["x"][0] == np.float32(x) and un["y"][0] == 0.0


def test_image_bounds_known_answers(orbref):
    K = (500.0, 500.0, 320.0, 240.0)
    assert orbref.image_bounds(K, (0.0, 0.3, 0.0, 0.0), 640, 480) == [0, 0, 640, 480]
    # barrel distortion (k1 < 0) pushes the undistorted corners outwards, symmetrically for a centred camera
    b = orbref.image_bounds(K, (-0.2, 0.0, 0.0, 0.0), 640, 480)
    assert b[0] < 0 and b[1] < 0 and b[2] > 640 and b[3] > 480
    assert b[0] == -(b[2] - 640) and b[1] == -(b[3] - 480)
