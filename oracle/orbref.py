"""ctypes binding of the CPU oracle (oracle/liborbref.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, the smoke check and the cpu_baseline leg of bench.py import this module; the product
package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liborbref.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("orb_ref.cpp", "frame_ref.cpp", "bow_ref.cpp", "orb_ref.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "liborbref.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
        L.orbref_create.restype = vp
        L.orbref_create.argtypes = [i32, f32, i32, i32, i32]
        L.orbref_destroy.argtypes = [vp]
        L.orbref_levels.argtypes = [vp]
        L.orbref_scale_factor.restype = f32
        L.orbref_scale_factor.argtypes = [vp]
        L.orbref_table.argtypes = [vp, i32, vp]
        L.orbref_extract.argtypes = [vp, vp, i32, i32, C.c_ssize_t, vp, vp, i32, vp]
        L.orbref_level_size.argtypes = [vp, i32, vp, vp]
        L.orbref_plane.argtypes = [vp, i32, i32, vp]
        L.orbref_level_keypoints.argtypes = [vp, i32, vp, i32]
        L.orbref_level_grid.argtypes = [vp, i32, vp]
        L.orbref_level_candidates.argtypes = [vp, i32, vp, vp, i32]
        L.orbref_level_retain.argtypes = [vp, i32, vp]
        L.orbref_cvround.argtypes = [C.c_double]
        L.orbref_fast_atan2.restype = f32
        L.orbref_fast_atan2.argtypes = [f32, f32]
        L.orbref_resize_linear.argtypes = [vp, i32, i32, C.c_ssize_t, vp, i32, i32, C.c_ssize_t]
        L.orbref_reflect101.argtypes = [i32, i32]
        L.orbref_fast.argtypes = [vp, i32, i32, C.c_ssize_t, i32, vp, i32]
        L.orbref_fast_measure.argtypes = [vp, C.c_ssize_t]
        L.orbref_retain_best.argtypes = [vp, i32, i32, vp]
        L.orbref_nth_prefix.argtypes = [vp, i32, i32, vp]
        L.orbref_gauss7.argtypes = [vp, i32, i32, vp]
        L.orbref_set_gauss_profile.argtypes = [i32]
        L.orbref_set_gauss_profile.restype = None
        L.orbref_ic_angle.restype = f32
        L.orbref_ic_angle.argtypes = [vp, C.c_ssize_t]
        L.orbref_descriptor.argtypes = [vp, C.c_ssize_t, f32, vp]
        L.orbref_cosf.restype = f32
        L.orbref_cosf.argtypes = [f32]
        L.orbref_sinf.restype = f32
        L.orbref_sinf.argtypes = [f32]
        L.orbref_hamming.argtypes = [vp, vp]
        L.orbref_knn2.argtypes = [vp, i32, vp, i32, vp, vp, vp]
        L.orbref_knn2_indexed.argtypes = [vp, vp, vp, i32, vp, vp, vp]
        L.orbref_ratio_accept.argtypes = [i32, i32, f32, i32]
        L.orbref_three_maxima.argtypes = [vp, i32, vp]
        L.orbref_rot_bin.argtypes = [f32, f32]
        L.orbref_frame_grid.restype = None
        L.orbref_frame_grid.argtypes = [vp, i32, vp, vp, vp]
        L.orbref_features_in_area.argtypes = [vp, i32, vp, f32, f32, f32, i32, i32, vp, i32]
        L.orbref_search_for_initialization.argtypes = [vp, vp, i32, vp, vp, i32, vp, vp, i32, f32, i32, vp]
        L.orbref_vocabulary_create.restype = vp
        L.orbref_vocabulary_create.argtypes = [i32, vp, vp, vp, i32, i32, i32]
        L.orbref_vocabulary_destroy.argtypes = [vp]
        L.orbref_vocabulary_destroy.restype = None
        L.orbref_bow_transform_feature.argtypes = [vp, vp, i32, vp, vp, vp]
        L.orbref_bow_transform_feature.restype = None
        L.orbref_bow_transform.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]
        L.orbref_search_by_bow_kf_frame.argtypes = [vp, vp, i32, vp, vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, i32, f32, i32, vp]
        L.orbref_search_by_bow_kf_kf.argtypes = [vp, vp, i32, vp, vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, i32, f32, i32, vp]
        L.orbref_bow_score_l1.restype = C.c_double
        L.orbref_bow_score_l1.argtypes = [vp, vp, i32, vp, vp, i32]
        L.orbref_distinctive_descriptor.argtypes = [vp, i32, vp]
        L.orbref_window_search.argtypes = [vp, vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, f32, i32, vp]
        L.orbref_undistort_keypoints.argtypes = [vp, i32, vp, vp, vp]
        L.orbref_undistort_keypoints.restype = None
        L.orbref_image_bounds.argtypes = [vp, vp, i32, i32, vp]
        L.orbref_image_bounds.restype = None
        L.orbref_sbp_local_map.argtypes = [vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, f32, f32, vp, vp]
        L.orbref_sbp_sim3.argtypes = [vp, vp, i32, vp, f32, f32, f32, f32, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp]
        L.orbref_sbp_frames.argtypes = [vp, vp, i32, vp, vp, vp, f32, f32, f32, f32, vp, vp, i32, vp, vp, i32, f32, vp, vp]
        L.orbref_sbp_last_frame.argtypes = [vp, vp, i32, vp, vp, f32, f32, f32, f32, vp, vp, vp, vp, i32, vp, vp, f32, i32, vp, vp]
        L.orbref_sbp_keyframe.argtypes = [vp, vp, i32, vp, vp, f32, f32, f32, f32, vp, vp, i32, i32, vp, vp, vp, vp, vp, f32,
                                          i32, i32, vp, vp]
        L.orbref_search_for_triangulation.argtypes = [vp, vp, i32, vp, vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, i32, vp, vp, i32, vp]
        L.orbref_is_in_frustum.argtypes = [vp, f32, f32, f32, f32, vp, vp, i32, i32, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp]
        fuse_args = [vp, vp, i32, vp, f32, f32, f32, f32, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp]
        L.orbref_fuse_pose.argtypes = fuse_args
        L.orbref_fuse_sim3.argtypes = fuse_args
        L.orbref_search_by_sim3.argtypes = [vp, vp, i32, vp, vp, i32, vp, f32, f32, f32, f32, vp, i32, vp, i32, vp, vp, f32, vp, vp,
                                            f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class RefExtractor:
    """Mirror of ORB_SLAM::ORBextractor (include/ORBextractor.h:32-77) over the oracle."""
    HARRIS_SCORE, FAST_SCORE = 0, 1

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, scoreType=1, fastTh=20):
        self._h = lib().orbref_create(nfeatures, scaleFactor, nlevels, scoreType, fastTh)
        if not self._h:
            raise ValueError("bad extractor parameters")
        self.nfeatures, self.nlevels = nfeatures, nlevels

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orbref_destroy(self._h)
            self._h = None

    def GetLevels(self):
        return lib().orbref_levels(self._h)

    def GetScaleFactor(self):
        return lib().orbref_scale_factor(self._h)

    def table(self, which):
        dt = np.float32 if which in (1, 2) else np.int32
        out = np.zeros(64, dtype=dt)
        n = lib().orbref_table(self._h, which, _p(out))
        return out[:n].copy()

    def __call__(self, image):
        """Returns (keypoints[N] structured, descriptors[N,32] u8)."""
        image = np.ascontiguousarray(image, dtype=np.uint8)
        rows, cols = image.shape if image.ndim == 2 else (0, 0)
        cap = self.nfeatures
        kp = np.zeros(cap, dtype=KP_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        n = C.c_int(0)
        rc = lib().orbref_extract(self._h, _p(image), rows, cols, image.strides[0] if image.ndim == 2 else 0,
                                  _p(kp), _p(desc), cap, C.byref(n))
        if rc != 0:
            raise RuntimeError("orbref_extract failed (%d)" % rc)
        return kp[:n.value].copy(), desc[:n.value].copy()

    # ---- stage taps ----
    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        lib().orbref_level_size(self._h, level, C.byref(w), C.byref(h))
        return w.value, h.value

    def plane(self, level, blurred=False):
        w, h = self.level_size(level)
        out = np.zeros((h + 32, w + 32), dtype=np.uint8)
        n = lib().orbref_plane(self._h, level, int(blurred), _p(out))
        return out if n else None

    def level_keypoints(self, level):
        out = np.zeros((self.nfeatures * 2 + 16, 3), dtype=np.int32)
        n = lib().orbref_level_keypoints(self._h, level, _p(out), out.shape[0])
        return out[:n].copy()

    def level_grid(self, level):
        out = np.zeros(5, dtype=np.int32)
        lib().orbref_level_grid(self._h, level, _p(out))
        return dict(cols=int(out[0]), rows=int(out[1]), cellW=int(out[2]), cellH=int(out[3]), nfeaturesCell=int(out[4]))

    def level_candidates(self, level):
        g = self.level_grid(level)
        w, h = self.level_size(level)
        counts = np.zeros(g["cols"] * g["rows"], dtype=np.int32)
        cap = w * h // 4 + 64
        xyr = np.zeros((cap, 3), dtype=np.int32)
        n = lib().orbref_level_candidates(self._h, level, _p(counts), _p(xyr), cap)
        return counts, xyr[:n].copy()

    def level_retain(self, level):
        g = self.level_grid(level)
        out = np.zeros(g["cols"] * g["rows"], dtype=np.int32)
        lib().orbref_level_retain(self._h, level, _p(out))
        return out


def hamming(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    return lib().orbref_hamming(_p(a), _p(b))


def knn2(q, t):
    q = np.ascontiguousarray(q, dtype=np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, dtype=np.uint8).reshape(-1, 32)
    nq, nt = q.shape[0], t.shape[0]
    idx = np.zeros(nq, np.int32)
    d1 = np.zeros(nq, np.int32)
    d2 = np.zeros(nq, np.int32)
    lib().orbref_knn2(_p(q), nq, _p(t), nt, _p(idx), _p(d1), _p(d2))
    return idx, d1, d2


# ---- Frame feature grid and windowed matchers (oracle/frame_ref.cpp) ----------------------------
FRAME_GRID_COLS, FRAME_GRID_ROWS = 64, 48
INT_MAX = 2**31 - 1


def _bounds(b):
    return np.ascontiguousarray(b, dtype=np.int32).reshape(4)


def frame_grid(kp, bounds):
    """CSR form of Frame::mGrid: (cell_start[64*48+1], items[...]) with cell = ix*48 + iy."""
    kp = np.ascontiguousarray(kp, dtype=KP_DTYPE)
    b = _bounds(bounds)
    start = np.zeros(FRAME_GRID_COLS * FRAME_GRID_ROWS + 1, np.int32)
    items = np.zeros(max(len(kp), 1), np.int32)
    lib().orbref_frame_grid(_p(kp), len(kp), _p(b), _p(start), _p(items))
    return start, items[:start[-1]].copy()


def features_in_area(kp, bounds, x, y, r, min_level=-1, max_level=-1):
    kp = np.ascontiguousarray(kp, dtype=KP_DTYPE)
    b = _bounds(bounds)
    out = np.zeros(max(len(kp), 1), np.int32)
    n = lib().orbref_features_in_area(_p(kp), len(kp), _p(b), x, y, r, min_level, max_level, _p(out), len(out))
    return out[:n].copy()


def search_for_initialization(kp1, desc1, kp2, desc2, bounds, prev_matched, window_size=100, nnratio=0.9,
                              check_orientation=True):
    """Returns (nmatches, vnMatches12, updated vbPrevMatched)."""
    kp1 = np.ascontiguousarray(kp1, dtype=KP_DTYPE)
    kp2 = np.ascontiguousarray(kp2, dtype=KP_DTYPE)
    desc1 = np.ascontiguousarray(desc1, dtype=np.uint8).reshape(-1, 32)
    desc2 = np.ascontiguousarray(desc2, dtype=np.uint8).reshape(-1, 32)
    b = _bounds(bounds)
    prev = np.array(prev_matched, dtype=np.float32).reshape(-1, 2).copy()
    m12 = np.zeros(max(len(kp1), 1), np.int32)
    n = lib().orbref_search_for_initialization(_p(kp1), _p(desc1), len(kp1), _p(kp2), _p(desc2), len(kp2), _p(b),
                                               _p(prev), window_size, nnratio, int(check_orientation), _p(m12))
    return n, m12[:len(kp1)].copy(), prev


def window_search(kp1, desc1, valid1, kp2, desc2, bounds, window_size, min_level=-1, max_level=INT_MAX, nnratio=0.6,
                  check_orientation=True):
    """Returns (nmatches, vnMatches21)."""
    kp1 = np.ascontiguousarray(kp1, dtype=KP_DTYPE)
    kp2 = np.ascontiguousarray(kp2, dtype=KP_DTYPE)
    desc1 = np.ascontiguousarray(desc1, dtype=np.uint8).reshape(-1, 32)
    desc2 = np.ascontiguousarray(desc2, dtype=np.uint8).reshape(-1, 32)
    valid1 = np.ascontiguousarray(valid1, dtype=np.uint8)
    b = _bounds(bounds)
    m21 = np.zeros(max(len(kp2), 1), np.int32)
    n = lib().orbref_window_search(_p(kp1), _p(desc1), len(kp1), _p(valid1), _p(kp2), _p(desc2), len(kp2), _p(b),
                                   window_size, min_level, max_level, nnratio, int(check_orientation), _p(m21))
    return n, m21[:len(kp2)].copy()


PROBE_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("r", "<f4"), ("min_level", "<i4"), ("max_level", "<i4"),
                        ("angle", "<f4"), ("valid", "<i4"), ("reserved", "<i4")])


def _frame(kp, desc):
    return (np.ascontiguousarray(kp, dtype=KP_DTYPE), np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, 32))


def _u8(a, n):
    return np.zeros(max(n, 1), np.uint8) if a is None else np.ascontiguousarray(a, dtype=np.uint8)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def sbp_local_map(kp, desc, bounds, scale_factors, taken, track_in_view, bad, level, view_cos, proj_x, proj_y, mp_desc,
                  th, nnratio):
    """SearchByProjection(Frame&, vector<MapPoint*>&, th): (nmatches, match_of_feature, probes)."""
    kp, desc = _frame(kp, desc)
    nmp = len(level)
    out = np.zeros(max(len(kp), 1), np.int32)
    probes = np.zeros(max(nmp, 1), PROBE_DTYPE)
    sf, b = _f32(scale_factors), _bounds(bounds)
    args = [_u8(track_in_view, nmp), _u8(bad, nmp), np.ascontiguousarray(level, dtype=np.int32), _f32(view_cos),
            _f32(proj_x), _f32(proj_y), np.ascontiguousarray(mp_desc, dtype=np.uint8)]
    tk = _u8(taken, len(kp))
    n = lib().orbref_sbp_local_map(_p(kp), _p(desc), len(kp), _p(b), _p(sf), _p(tk), nmp, *[_p(a) for a in args], th, nnratio,
                                   _p(out), _p(probes))
    return n, out[:len(kp)].copy(), probes[:nmp].copy()


def sbp_sim3(kp, desc, bounds, K, scale_factors, Scw, matched, skip, pos, normal, min_dist, max_dist, mp_desc, th):
    """SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th): (nmatches, match_of_feature, probes)."""
    kp, desc = _frame(kp, desc)
    npts = len(skip)
    out = np.zeros(max(len(kp), 1), np.int32)
    probes = np.zeros(max(npts, 1), PROBE_DTYPE)
    sf, b, S = _f32(scale_factors), _bounds(bounds), _f32(Scw).reshape(16)
    mt = _u8(matched, len(kp))
    args = [_u8(skip, npts), _f32(pos), _f32(normal), _f32(min_dist), _f32(max_dist), np.ascontiguousarray(mp_desc, dtype=np.uint8)]
    n = lib().orbref_sbp_sim3(_p(kp), _p(desc), len(kp), _p(b), K[0], K[1], K[2], K[3], _p(sf), len(sf), _p(S), _p(mt), npts,
                              *[_p(a) for a in args], int(th), _p(out), _p(probes))
    return n, out[:len(kp)].copy(), probes[:npts].copy()


def fuse(kp, desc, bounds, K, scale_factors, T, skip, pos, normal, min_dist, max_dist, mp_desc, th, sim3=False):
    """Fuse(KeyFrame*, vector<MapPoint*>&, th) (T = the keyframe's Tcw) or, with sim3=True, Fuse(KeyFrame*, Scw,
    vpPoints, th) (T = Scw): (nFused, best_idx, best_dist, probes) per map point."""
    kp, desc = _frame(kp, desc)
    npts = len(skip)
    bi = np.zeros(max(npts, 1), np.int32)
    bd = np.zeros(max(npts, 1), np.int32)
    probes = np.zeros(max(npts, 1), PROBE_DTYPE)
    sf, b, Tm = _f32(scale_factors), _bounds(bounds), _f32(T).reshape(16)
    args = [_u8(skip, npts), _f32(pos), _f32(normal), _f32(min_dist), _f32(max_dist), np.ascontiguousarray(mp_desc, dtype=np.uint8)]
    f = lib().orbref_fuse_sim3 if sim3 else lib().orbref_fuse_pose
    n = f(_p(kp), _p(desc), len(kp), _p(b), K[0], K[1], K[2], K[3], _p(sf), len(sf), _p(Tm), npts, *[_p(a) for a in args],
          float(th), _p(bi), _p(bd), _p(probes))
    return n, bi[:npts].copy(), bd[:npts].copy(), probes[:npts].copy()


def search_by_sim3(kp1, desc1, kp2, desc2, bounds, K, scale_factors, T1w, T2w, s12, R12, t12, th, skip1, pos1, min1, max1,
                   mp_desc1, skip2, pos2, min2, max2, mp_desc2):
    """SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th): (nFound, matches12 [n1], vnMatch1 [n1], vnMatch2 [n2],
    probes1 [n1], probes2 [n2])."""
    kp1, desc1 = _frame(kp1, desc1)
    kp2, desc2 = _frame(kp2, desc2)
    n1, n2 = len(kp1), len(kp2)
    assert len(skip1) == n1 and len(skip2) == n2
    m1 = np.zeros(max(n1, 1), np.int32)
    m2 = np.zeros(max(n2, 1), np.int32)
    m12 = np.zeros(max(n1, 1), np.int32)
    pr1 = np.zeros(max(n1, 1), PROBE_DTYPE)
    pr2 = np.zeros(max(n2, 1), PROBE_DTYPE)
    sf, b = _f32(scale_factors), _bounds(bounds)
    A = [_f32(T1w).reshape(16), _f32(T2w).reshape(16)]
    R, t = _f32(R12).reshape(9), _f32(t12).reshape(3)
    side1 = [_u8(skip1, n1), _f32(pos1), _f32(min1), _f32(max1), np.ascontiguousarray(mp_desc1, dtype=np.uint8)]
    side2 = [_u8(skip2, n2), _f32(pos2), _f32(min2), _f32(max2), np.ascontiguousarray(mp_desc2, dtype=np.uint8)]
    n = lib().orbref_search_by_sim3(_p(kp1), _p(desc1), n1, _p(kp2), _p(desc2), n2, _p(b), K[0], K[1], K[2], K[3], _p(sf), len(sf),
                                    _p(sf), len(sf), _p(A[0]), _p(A[1]), float(s12), _p(R), _p(t), float(th),
                                    *[_p(a) for a in side1], *[_p(a) for a in side2], _p(m1), _p(m2), _p(m12), _p(pr1), _p(pr2))
    return n, m12[:n1].copy(), m1[:n1].copy(), m2[:n2].copy(), pr1[:n1].copy(), pr2[:n2].copy()


def sbp_frames(kp1, desc1, skip1, pos1, Tc2w, K, kp2, desc2, bounds, taken2, window_size, nnratio):
    """SearchByProjection(Frame &F1, Frame &F2, windowSize, vpMapPointMatches2): (nmatches, match_of_feature2, probes)."""
    kp1, desc1 = _frame(kp1, desc1)
    kp2, desc2 = _frame(kp2, desc2)
    out = np.zeros(max(len(kp2), 1), np.int32)
    probes = np.zeros(max(len(kp1), 1), PROBE_DTYPE)
    b, T = _bounds(bounds), _f32(Tc2w).reshape(16)
    sk, ps, tk = _u8(skip1, len(kp1)), _f32(pos1), _u8(taken2, len(kp2))
    n = lib().orbref_sbp_frames(_p(kp1), _p(desc1), len(kp1), _p(sk), _p(ps), _p(T), K[0], K[1], K[2], K[3], _p(kp2), _p(desc2),
                                len(kp2), _p(b), _p(tk), int(window_size), nnratio, _p(out), _p(probes))
    return n, out[:len(kp2)].copy(), probes[:len(kp1)].copy()


def sbp_last_frame(kp_cur, desc_cur, bounds, taken_cur, K, Tcw, scale_factors, kp_last, desc_last, skip_last, pos_last, th,
                   check_orientation):
    """SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th): (nmatches, match_of_feature, probes)."""
    kp_cur, desc_cur = _frame(kp_cur, desc_cur)
    kp_last, desc_last = _frame(kp_last, desc_last)
    out = np.zeros(max(len(kp_cur), 1), np.int32)
    probes = np.zeros(max(len(kp_last), 1), PROBE_DTYPE)
    b, T, sf = _bounds(bounds), _f32(Tcw).reshape(16), _f32(scale_factors)
    tk, sk, ps = _u8(taken_cur, len(kp_cur)), _u8(skip_last, len(kp_last)), _f32(pos_last)
    n = lib().orbref_sbp_last_frame(_p(kp_cur), _p(desc_cur), len(kp_cur), _p(b), _p(tk), K[0], K[1], K[2], K[3], _p(T), _p(sf),
                                    _p(kp_last), _p(desc_last), len(kp_last), _p(sk), _p(ps), th, int(check_orientation),
                                    _p(out), _p(probes))
    return n, out[:len(kp_cur)].copy(), probes[:len(kp_last)].copy()


def sbp_keyframe(kp_cur, desc_cur, bounds, taken_cur, K, Tcw, scale_factors, skip_kf, pos, min_dist, mp_desc, kf_angle, th,
                 orb_dist, check_orientation):
    """SearchByProjection(Frame &CurrentFrame, KeyFrame*, sAlreadyFound, th, ORBdist): (nmatches, match_of_feature, probes)."""
    kp_cur, desc_cur = _frame(kp_cur, desc_cur)
    nkf = len(skip_kf)
    out = np.zeros(max(len(kp_cur), 1), np.int32)
    probes = np.zeros(max(nkf, 1), PROBE_DTYPE)
    b, T, sf = _bounds(bounds), _f32(Tcw).reshape(16), _f32(scale_factors)
    tk = _u8(taken_cur, len(kp_cur))
    args = [_u8(skip_kf, nkf), _f32(pos), _f32(min_dist), np.ascontiguousarray(mp_desc, dtype=np.uint8), _f32(kf_angle)]
    n = lib().orbref_sbp_keyframe(_p(kp_cur), _p(desc_cur), len(kp_cur), _p(b), _p(tk), K[0], K[1], K[2], K[3], _p(T), _p(sf),
                                  len(sf), nkf, *[_p(a) for a in args], th, int(orb_dist), int(check_orientation), _p(out),
                                  _p(probes))
    return n, out[:len(kp_cur)].copy(), probes[:nkf].copy()


def undistort_keypoints(kp, K, dist):
    """Frame::UndistortKeyPoints: mvKeysUn from mvKeys, K = (fx, fy, cx, cy), dist = (k1, k2, p1, p2)."""
    kp = np.ascontiguousarray(kp, dtype=KP_DTYPE)
    out = np.zeros(max(len(kp), 1), KP_DTYPE)
    k, d = _f32(K), _f32(dist)
    lib().orbref_undistort_keypoints(_p(kp), len(kp), _p(k), _p(d), _p(out))
    return out[:len(kp)].copy()


def image_bounds(K, dist, cols, rows):
    """Frame::ComputeImageBounds: [mnMinX, mnMinY, mnMaxX, mnMaxY]."""
    b = np.zeros(4, np.int32)
    k, d = _f32(K), _f32(dist)
    lib().orbref_image_bounds(_p(k), _p(d), int(cols), int(rows), _p(b))
    return [int(v) for v in b]


def distinctive_descriptor(desc):
    """MapPoint::ComputeDistinctiveDescriptors on one set of observations: (BestIdx, BestMedian)."""
    desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, 32)
    med = C.c_int32(0)
    idx = lib().orbref_distinctive_descriptor(_p(desc), len(desc), C.byref(med))
    return idx, med.value


# ---- Bag-of-words transform (oracle/bow_ref.cpp) --------------------------------------------------
class RefVocabulary:
    """The vocabulary tree of Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h over the oracle.
    parent[nnodes] (node 0 = root), desc[nnodes, 32], weight[nnodes] (float64)."""

    def __init__(self, parent, desc, weight, L, weighting=0, scoring=0):
        self.parent = np.ascontiguousarray(parent, dtype=np.int32)
        self.desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, 32)
        self.weight = np.ascontiguousarray(weight, dtype=np.float64)
        self.L = L
        self._h = lib().orbref_vocabulary_create(len(self.parent), _p(self.parent), _p(self.desc), _p(self.weight), L,
                                                 weighting, scoring)
        if not self._h:
            raise ValueError("bad vocabulary")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orbref_vocabulary_destroy(self._h)
            self._h = None

    def transform_features(self, features, levelsup=4):
        """Per feature: (word id, node id at level L - levelsup, weight)."""
        f = np.ascontiguousarray(features, dtype=np.uint8).reshape(-1, 32)
        word = np.zeros(len(f), np.int32)
        node = np.zeros(len(f), np.int32)
        w = np.zeros(len(f), np.float64)
        wi, ni, we = C.c_int32(), C.c_int32(), C.c_double()
        for i in range(len(f)):
            lib().orbref_bow_transform_feature(self._h, _p(f[i]), levelsup, C.byref(wi), C.byref(ni), C.byref(we))
            word[i], node[i], w[i] = wi.value, ni.value, we.value
        return word, node, w

    def transform(self, features, levelsup=4):
        """(BowVector as (words, values), FeatureVector as (nodes, start, feature indices))."""
        f = np.ascontiguousarray(features, dtype=np.uint8).reshape(-1, 32)
        n = len(f)
        bw, bv = np.zeros(n + 1, np.int32), np.zeros(n + 1, np.float64)
        fn, fs, ff = np.zeros(n + 1, np.int32), np.zeros(n + 2, np.int32), np.zeros(n + 1, np.int32)
        nw, nn = C.c_int32(), C.c_int32()
        lib().orbref_bow_transform(self._h, _p(f), n, levelsup, _p(bw), _p(bv), C.byref(nw), _p(fn), _p(fs), _p(ff),
                                   C.byref(nn))
        return (bw[:nw.value].copy(), bv[:nw.value].copy()), (fn[:nn.value].copy(), fs[:nn.value + 1].copy(),
                                                                ff[:fs[nn.value]].copy())


def bow_score_l1(a, b):
    w1, v1 = np.ascontiguousarray(a[0], np.int32), np.ascontiguousarray(a[1], np.float64)
    w2, v2 = np.ascontiguousarray(b[0], np.int32), np.ascontiguousarray(b[1], np.float64)
    return lib().orbref_bow_score_l1(_p(w1), _p(v1), len(w1), _p(w2), _p(v2), len(w2))


def _fv_arrays(fv):
    nodes, start, feat = fv
    return (np.ascontiguousarray(nodes, np.int32), np.ascontiguousarray(start, np.int32),
            np.ascontiguousarray(feat if len(feat) else np.zeros(1), np.int32))


def search_by_bow_kf_frame(kpKF, descKF, validKF, fvKF, kpF, descF, fvF, nnratio=0.7, check_orientation=True):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...): (nmatches, matchesF[nF] = keyframe index or -1)."""
    kpKF, kpF = np.ascontiguousarray(kpKF, KP_DTYPE), np.ascontiguousarray(kpF, KP_DTYPE)
    descKF = np.ascontiguousarray(descKF, np.uint8).reshape(-1, 32)
    descF = np.ascontiguousarray(descF, np.uint8).reshape(-1, 32)
    validKF = np.ascontiguousarray(validKF, np.uint8)
    a, b = _fv_arrays(fvKF), _fv_arrays(fvF)
    out = np.full(max(len(kpF), 1), -1, np.int32)
    n = lib().orbref_search_by_bow_kf_frame(_p(kpKF), _p(descKF), len(kpKF), _p(validKF), _p(a[0]), _p(a[1]), _p(a[2]),
                                            len(a[0]), _p(kpF), _p(descF), len(kpF), _p(b[0]), _p(b[1]), _p(b[2]), len(b[0]),
                                            nnratio, int(check_orientation), _p(out))
    return n, out[:len(kpF)].copy()


def search_by_bow_kf_kf(kp1, desc1, valid1, fv1, kp2, desc2, valid2, fv2, nnratio=0.8, check_orientation=True):
    """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, ...): (nmatches, matches12[n1] = index in keyframe 2 or -1)."""
    kp1, kp2 = np.ascontiguousarray(kp1, KP_DTYPE), np.ascontiguousarray(kp2, KP_DTYPE)
    desc1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32)
    desc2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
    valid1, valid2 = np.ascontiguousarray(valid1, np.uint8), np.ascontiguousarray(valid2, np.uint8)
    a, b = _fv_arrays(fv1), _fv_arrays(fv2)
    out = np.full(max(len(kp1), 1), -1, np.int32)
    n = lib().orbref_search_by_bow_kf_kf(_p(kp1), _p(desc1), len(kp1), _p(valid1), _p(a[0]), _p(a[1]), _p(a[2]), len(a[0]),
                                         _p(kp2), _p(desc2), len(kp2), _p(valid2), _p(b[0]), _p(b[1]), _p(b[2]), len(b[0]),
                                         nnratio, int(check_orientation), _p(out))
    return n, out[:len(kp1)].copy()


def search_for_triangulation(kp1, desc1, free1, fv1, kp2, desc2, free2, fv2, F12, level_sigma2, check_orientation=True):
    """ORBmatcher::SearchForTriangulation: (nmatches, vMatches12[n1] = index in keyframe 2 or -1).  free1 / free2:
    the feature holds no map point; F12: 3x3; level_sigma2: mvLevelSigma2 of keyframe 2."""
    kp1, kp2 = np.ascontiguousarray(kp1, KP_DTYPE), np.ascontiguousarray(kp2, KP_DTYPE)
    desc1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32)
    desc2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
    free1, free2 = np.ascontiguousarray(free1, np.uint8), np.ascontiguousarray(free2, np.uint8)
    a, b = _fv_arrays(fv1), _fv_arrays(fv2)
    F, s2 = _f32(F12).reshape(9), _f32(level_sigma2)
    out = np.full(max(len(kp1), 1), -1, np.int32)
    n = lib().orbref_search_for_triangulation(_p(kp1), _p(desc1), len(kp1), _p(free1), _p(a[0]), _p(a[1]), _p(a[2]), len(a[0]),
                                              _p(kp2), _p(desc2), len(kp2), _p(free2), _p(b[0]), _p(b[1]), _p(b[2]), len(b[0]),
                                              _p(F), _p(s2), int(check_orientation), _p(out))
    return n, out[:len(kp1)].copy()


def is_in_frustum(bounds, K, Tcw, scale_factors, pos, normal, min_dist, max_dist, viewing_cos_limit=0.5):
    """Frame::isInFrustum for every map point: (in_view u8, proj_x, proj_y, level i32, view_cos)."""
    n = len(min_dist)
    b, sf, T = _bounds(bounds), _f32(scale_factors), _f32(Tcw).reshape(16)
    inv = np.zeros(max(n, 1), np.uint8)
    px, py, vc = (np.zeros(max(n, 1), np.float32) for _ in range(3))
    lv = np.zeros(max(n, 1), np.int32)
    a = [_f32(pos), _f32(normal), _f32(min_dist), _f32(max_dist)]
    lib().orbref_is_in_frustum(_p(b), K[0], K[1], K[2], K[3], _p(T), _p(sf), len(sf), n, *[_p(x) for x in a],
                               float(viewing_cos_limit), _p(inv), _p(px), _p(py), _p(lv), _p(vc))
    return inv[:n].copy(), px[:n].copy(), py[:n].copy(), lv[:n].copy(), vc[:n].copy()
