"""Host-side mirror of the accelerated part of ORB_SLAM::ORBmatcher (include/ORBmatcher.h:39-109):
DescriptorDistance, the best/second-best scan with the ratio test, and the static thresholds.
"""
import ctypes as C

import numpy as np

from . import _lib

INT_MAX = 2147483647


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class ORBmatcher:
    TH_HIGH = 100      # src/ORBmatcher.cc:42
    TH_LOW = 20        # :44
    HISTO_LENGTH = 30  # :45

    def __init__(self, nnratio=0.6, checkOri=True, device=0):
        self.mfNNratio = float(np.float32(nnratio))
        self.mbCheckOrientation = bool(checkOri)
        self._L = _lib.lib()
        self._h = C.c_void_p()
        _lib.check(self._L.orbhip_matcher_create(C.byref(self._h), device))

        self.device = device
        self._async = False
        self._tstream = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.orbhip_matcher_destroy(self._h)
            self._h = C.c_void_p()

    # ---- stream discipline of the batched device methods ---------------------------------------------
    # The library works on the matcher's own stream.  Blocking form (default): wait for everything torch has
    # queued, call, wait for the matcher.  Asynchronous form (after reserve_candidates): the matcher's stream
    # waits for torch's current stream on the device, the call is queued, torch's current stream then waits for
    # the matcher's stream on the device; the host never waits until the caller synchronises.
    def reserve_candidates(self, entries):
        """orbhip_matcher_reserve_candidates: room for `entries` candidate records (4 bytes each) of the windowed and
        projection matchers; with a reservation in place they no longer read anything back and do not block.  A
        call whose candidate lists need more makes the next synchronize() raise with the count needed.  0 returns
        to the blocking form."""
        _lib.check(self._L.orbhip_matcher_reserve_candidates(self._h, int(entries)))
        self._async = entries > 0

    def set_scan_kernel(self, kind):
        """orbhip_matcher_set_scan_kernel: "matrix" (4-bit matrix-core scan, the default) or "popcount"."""
        _lib.check(self._L.orbhip_matcher_set_scan_kernel(self._h, {"matrix": 0, "popcount": 1}[kind]))

    def torch_stream(self):
        import torch
        if self._tstream is None:
            self._tstream = torch.cuda.ExternalStream(self._L.orbhip_matcher_stream(self._h), device="cuda:%d" % self.device)
        return self._tstream

    def _begin(self):
        import torch
        if self._async:
            self.torch_stream().wait_stream(torch.cuda.current_stream(self.device))
        else:
            torch.cuda.synchronize()

    def _end(self):
        import torch
        if self._async:
            torch.cuda.current_stream(self.device).wait_stream(self.torch_stream())
        else:
            self.synchronize()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def DescriptorDistance(a, b):
        """static int DescriptorDistance(const cv::Mat&, const cv::Mat&), include/ORBmatcher.h:46."""
        a = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1)
        b = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1)
        assert a.size == 32 and b.size == 32
        return _lib.lib().orbhip_hamming_distance(_p(a), _p(b))

    def knn2(self, queries, train):
        """queries [P, nq, 32] or [nq, 32]; train [P, nt, 32] or [nt, 32] (host arrays).
        Returns (idx, d1, d2) int32 arrays of shape [P, nq] or [nq]."""
        q = np.ascontiguousarray(queries, dtype=np.uint8)
        t = np.ascontiguousarray(train, dtype=np.uint8)
        single = q.ndim == 2
        if single:
            q, t = q[None], t[None]
        assert q.shape[0] == t.shape[0] and q.shape[2] == 32 and t.shape[2] == 32
        P, nq, nt = q.shape[0], q.shape[1], t.shape[1]
        idx = np.full((P, nq), -1, np.int32)
        d1 = np.full((P, nq), INT_MAX, np.int32)
        d2 = np.full((P, nq), INT_MAX, np.int32)
        _lib.check(self._L.orbhip_knn2(self._h, _p(q), nq, _p(t), nt, P, _p(idx), _p(d1), _p(d2)))
        if single:
            return idx[0], d1[0], d2[0]
        return idx, d1, d2

    def knn2_device(self, d_q, nq, q_pair_stride, d_t, nt, t_pair_stride, npairs, idx_base, d_idx, d_d1, d_d2):
        _lib.check(self._L.orbhip_knn2_device(self._h, d_q, nq, q_pair_stride, d_t, nt, t_pair_stride, npairs, idx_base,
                                              d_idx, d_d1, d_d2))

    def knn2_device_noalloc(self, d_q, nq, q_pair_stride, d_t, nt, t_pair_stride, npairs, idx_base, d_idx, d_d1, d_d2):
        """orbhip_knn2_device_noalloc: the same scan with its partial results in the GPU's fixed scratch
        array; makes no device allocation."""
        _lib.check(self._L.orbhip_knn2_device_noalloc(self._h, d_q, nq, q_pair_stride, d_t, nt, t_pair_stride, npairs,
                                                      idx_base, d_idx, d_d1, d_d2))

    def merge_device(self, d_parts, nshards, nq, d_idx, d_d1, d_d2):
        _lib.check(self._L.orbhip_knn2_merge_device(self._h, d_parts, nshards, nq, d_idx, d_d1, d_d2))

    def ratio_test_device(self, d_d1, d_d2, n, form, th, d_accept, ratio=None):
        r = self.mfNNratio if ratio is None else ratio
        _lib.check(self._L.orbhip_ratio_test_device(self._h, d_d1, d_d2, n, r, form, th, d_accept))

    def SearchForInitialization(self, F1, F2, vbPrevMatched, windowSize=10):
        """int SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched,
        vector<int> &vnMatches12, int windowSize=10) (include/ORBmatcher.h:76, src/ORBmatcher.cc:609-724)
        for every frame pair of two FrameBatch objects.  vbPrevMatched: float32 CUDA tensor
        [batch, cap, 2], updated in place.  Returns (nmatches [batch], vnMatches12 [batch, cap]) as
        int32 CUDA tensors."""
        import torch
        assert F1.batch == F2.batch and F1.cap == F2.cap
        assert vbPrevMatched.is_cuda and vbPrevMatched.dtype == torch.float32 and vbPrevMatched.is_contiguous()
        dev = F1.n.device
        m12 = torch.empty((F1.batch, F1.cap), dtype=torch.int32, device=dev)
        nm = torch.empty(F1.batch, dtype=torch.int32, device=dev)
        self._begin()
        _lib.check(self._L.orbhip_search_for_initialization_device(
            self._h, F1.kp.data_ptr(), F1.desc.data_ptr(), F1.n.data_ptr(), F2.kp.data_ptr(), F2.desc.data_ptr(),
            F2.n.data_ptr(), F2.cell_start.data_ptr(), F2.cell_items.data_ptr(), F1.batch, F1.cap, F2.bounds,
            vbPrevMatched.data_ptr(), int(windowSize), self.mfNNratio, int(self.mbCheckOrientation), m12.data_ptr(),
            nm.data_ptr()))
        self._end()
        return nm, m12

    def WindowSearch(self, F1, F2, windowSize, valid1, minScaleLevel=-1, maxScaleLevel=INT_MAX):
        """int WindowSearch(Frame &F1, Frame &F2, int windowSize, vector<MapPoint*> &vpMapPointMatches2,
        int minScaleLevel=-1, int maxScaleLevel=INT_MAX) (include/ORBmatcher.h:71, src/ORBmatcher.cc:420-527).
        valid1: uint8 CUDA tensor [batch, cap], non-zero where F1.mvpMapPoints[i] is a good map point.
        Returns (nmatches [batch], vnMatches21 [batch, cap]): vpMapPointMatches2[i2] =
        F1.mvpMapPoints[vnMatches21[i2]] where that is >= 0."""
        import torch
        assert F1.batch == F2.batch and F1.cap == F2.cap
        assert valid1.is_cuda and valid1.dtype == torch.uint8 and valid1.is_contiguous()
        dev = F1.n.device
        m21 = torch.empty((F1.batch, F1.cap), dtype=torch.int32, device=dev)
        nm = torch.empty(F1.batch, dtype=torch.int32, device=dev)
        self._begin()
        _lib.check(self._L.orbhip_window_search_device(
            self._h, F1.kp.data_ptr(), F1.desc.data_ptr(), F1.n.data_ptr(), valid1.data_ptr(), F2.kp.data_ptr(),
            F2.desc.data_ptr(), F2.n.data_ptr(), F2.cell_start.data_ptr(), F2.cell_items.data_ptr(), F1.batch, F1.cap,
            F2.bounds, int(windowSize), int(minScaleLevel), int(maxScaleLevel), self.mfNNratio,
            int(self.mbCheckOrientation), m21.data_ptr(), nm.data_ptr()))
        self._end()
        return nm, m21

    # acceptance rules of the SearchByProjection overloads (include/orbhip.h)
    PROJ_LOCAL_MAP, PROJ_RATIO, PROJ_BEST = 0, 1, 2

    def SearchByProjection(self, F, probes, probe_desc, nprobes, rule, taken=None, th_dist=TH_HIGH, check_orientation=None):
        """Search and assignment loops of the five ORBmatcher::SearchByProjection overloads
        (src/ORBmatcher.cc:52-128, :297-418, :530-605, :1518-1631, :1633-1757) on a batch of target frames.
        probes: uint8 CUDA tensor [batch, probe_cap, 32] of PROBE_DTYPE records (one per projected map point,
        in the order the reference walks them), probe_desc: uint8 [batch, probe_cap, 32], nprobes: int32 [batch],
        taken: uint8 [batch, cap], non-zero where the frame's feature already holds a map point.
        Returns (nmatches [batch], match_of_feature [batch, cap]: probe index assigned to the feature or -1)."""
        import torch
        assert probes.is_cuda and probes.dtype == torch.uint8 and probes.is_contiguous() and probes.shape[-1] == 32
        assert probe_desc.is_cuda and probe_desc.dtype == torch.uint8 and probe_desc.is_contiguous()
        assert probes.shape[0] == F.batch and probe_desc.shape[:2] == probes.shape[:2]
        assert nprobes.is_cuda and nprobes.dtype == torch.int32
        if taken is not None:
            assert taken.is_cuda and taken.dtype == torch.uint8 and taken.is_contiguous() and tuple(taken.shape) == (F.batch, F.cap)
        dev = F.n.device
        out = torch.empty((F.batch, F.cap), dtype=torch.int32, device=dev)
        nm = torch.empty(F.batch, dtype=torch.int32, device=dev)
        ori = self.mbCheckOrientation if check_orientation is None else check_orientation
        self._begin()
        _lib.check(self._L.orbhip_search_by_projection_device(
            self._h, probes.data_ptr(), probe_desc.data_ptr(), nprobes.data_ptr(), int(probes.shape[1]), F.kp.data_ptr(),
            F.desc.data_ptr(), F.n.data_ptr(), F.cell_start.data_ptr(), F.cell_items.data_ptr(), F.batch, F.cap, F.bounds,
            0 if taken is None else taken.data_ptr(), int(rule), self.mfNNratio, int(th_dist), int(ori), out.data_ptr(),
            nm.data_ptr()))
        self._end()
        return nm, out

    PROBES_LAST_FRAME, PROBES_KEYFRAME, PROBES_FRUSTUM, PROBES_FUSE, PROBES_WINDOW, PROBES_SIM3 = 0, 1, 2, 3, 4, 5

    @staticmethod
    def projections(poses, K, sim3=False):
        """One OrbProjection per 4x4 pose (Tcw, or a Sim3 Scw with sim3=True), as a uint8 CUDA tensor [len(poses), 76]:
        Rcw, tcw, Ow = -Rcw.t()*tcw and the intrinsics K = (fx, fy, cx, cy)."""
        import torch
        L = _lib.lib()
        arr = (_lib.OrbProjection * max(len(poses), 1))()
        for i, T in enumerate(poses):
            t = np.ascontiguousarray(T, dtype=np.float32).reshape(16)
            f = L.orbhip_projection_from_sim3 if sim3 else L.orbhip_projection_from_pose
            _lib.check(f(_p(t), float(K[0]), float(K[1]), float(K[2]), float(K[3]), C.byref(arr[i])))
        raw = np.frombuffer(bytes(arr), np.uint8).reshape(max(len(poses), 1), C.sizeof(_lib.OrbProjection))
        return torch.from_numpy(raw.copy()).cuda()

    @staticmethod
    def sim3_projections(s12, R12, t12, K):
        """orbhip_projection_sim3_pair: the two similarities of SearchBySim3 (src/ORBmatcher.cc:1290-1292) as
        OrbProjection records on the device, (to2, to1): to2 moves keyframe 1's points into camera 2 (sR21, t21), to1
        the other way (sR12, t12); K = (fx, fy, cx, cy) of keyframe 1."""
        import torch
        L = _lib.lib()
        to2, to1 = _lib.OrbProjection(), _lib.OrbProjection()
        R = np.ascontiguousarray(R12, np.float32).reshape(9)
        t = np.ascontiguousarray(t12, np.float32).reshape(3)
        _lib.check(L.orbhip_projection_sim3_pair(float(s12), _p(R), _p(t), float(K[0]), float(K[1]), float(K[2]), float(K[3]),
                                                 C.byref(to2), C.byref(to1)))
        rec = lambda a: torch.from_numpy(np.frombuffer(bytes(a), np.uint8).reshape(1, C.sizeof(_lib.OrbProjection)).copy()).cuda()  # noqa: E731
        return rec(to2), rec(to1)

    def BuildProbesSim3(self, cams_from, cams_to, bounds, scale_factors, pos, npoints, th, min_dist, max_dist, skip=None):
        """One direction of SearchBySim3 (src/ORBmatcher.cc:1330-1375, :1415-1460) on the device: the points go through
        their own keyframe's pose (cams_from: projections()) and then through the similarity (cams_to:
        sim3_projections()).  Other arguments and the result as for BuildProbes."""
        import torch
        B, pcap = int(pos.shape[0]), int(pos.shape[1])
        assert pos.is_cuda and pos.dtype == torch.float32 and pos.is_contiguous() and pos.shape[2] == 3
        sf = scale_factors if torch.is_tensor(scale_factors) else torch.from_numpy(np.ascontiguousarray(scale_factors, np.float32)).cuda()
        probes = torch.empty((B, max(pcap, 1), 32), dtype=torch.uint8, device=pos.device)
        self._begin()
        _lib.check(self._L.orbhip_build_probes_sim3_device(
            self._h, cams_from.data_ptr(), cams_to.data_ptr(), _lib.OrbFrameBounds(*map(int, bounds)), sf.data_ptr(), int(sf.numel()),
            pos.data_ptr(), min_dist.data_ptr(), max_dist.data_ptr(), 0 if skip is None else skip.data_ptr(), npoints.data_ptr(), B,
            max(pcap, 1), float(th), probes.data_ptr()))
        self._end()
        return probes

    @staticmethod
    def probes_from_tracked(proj_xy, view_cos, level, skip, th, scale_factors):
        """orbhip_probes_from_tracked (host): the probes of SearchByProjection(F, vpMapPoints, th) from the fields
        Frame::isInFrustum left in the map points.  numpy in, PROBE_DTYPE array out."""
        from . import PROBE_DTYPE
        n = len(level)
        xy = np.ascontiguousarray(proj_xy, np.float32).reshape(-1)
        vc = np.ascontiguousarray(view_cos, np.float32)
        lv = np.ascontiguousarray(level, np.int32)
        sk = np.ascontiguousarray(skip, np.uint8)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        out = np.zeros(max(n, 1), PROBE_DTYPE)
        _lib.check(_lib.lib().orbhip_probes_from_tracked(n, _p(xy), _p(vc), _p(lv), _p(sk), float(th), _p(sf), len(sf), _p(out)))
        return out[:n]

    def BuildProbes(self, kind, cams, bounds, scale_factors, pos, npoints, th, normal=None, min_dist=None, max_dist=None,
                    src_kp=None, skip=None, viewing_cos_limit=0.5):
        """The lines of the projection matchers between their pointer tests and GetFeaturesInArea, on the device: one
        OrbProbe per map point (kind: PROBES_LAST_FRAME src/ORBmatcher.cc:1531-1556, PROBES_KEYFRAME :1651-1690,
        PROBES_FRUSTUM src/Frame.cc:138-199 + :63-81, PROBES_FUSE :1054-1094 / :1180-1223 / :326-375).
        cams: projections(); pos float32 [batch, pcap, 3]; normal likewise; min_dist / max_dist float32 [batch, pcap];
        src_kp uint8 [batch, pcap, 28] keypoint records; skip uint8 [batch, pcap]; npoints int32 [batch]; CUDA tensors.
        Returns the probes as a uint8 tensor [batch, pcap, 32] (PROBE_DTYPE records)."""
        import torch
        B, pcap = int(pos.shape[0]), int(pos.shape[1])
        assert pos.is_cuda and pos.dtype == torch.float32 and pos.is_contiguous() and pos.shape[2] == 3
        sf = scale_factors if torch.is_tensor(scale_factors) else torch.from_numpy(np.ascontiguousarray(scale_factors, np.float32)).cuda()
        probes = torch.empty((B, max(pcap, 1), 32), dtype=torch.uint8, device=pos.device)
        ptr = lambda t: 0 if t is None else t.data_ptr()  # noqa: E731
        for t in (normal, min_dist, max_dist, src_kp, skip):
            assert t is None or (t.is_cuda and t.is_contiguous())
        self._begin()
        _lib.check(self._L.orbhip_build_probes_device(
            self._h, int(kind), cams.data_ptr(), _lib.OrbFrameBounds(*map(int, bounds)), sf.data_ptr(), int(sf.numel()),
            pos.data_ptr(), ptr(normal), ptr(min_dist), ptr(max_dist), ptr(src_kp), ptr(skip), npoints.data_ptr(), B, max(pcap, 1),
            float(th), float(viewing_cos_limit), probes.data_ptr()))
        self._end()
        return probes

    def BestInRadius(self, F, probes, probe_desc, nprobes):
        """The per-map-point search of ORBmatcher::Fuse (src/ORBmatcher.cc:1096-1124, :1225-1253) and
        ORBmatcher::SearchBySim3 (:1376-1404, :1461-1489) on a batch of keyframes: GetFeaturesInArea(u, v, r), the
        level test of the probe and the first candidate with the least distance; probes are independent.
        Arguments as for SearchByProjection.  Returns (best_idx [batch, probe_cap]: bestIdx or -1,
        best_dist [batch, probe_cap]: bestDist or INT_MAX)."""
        import torch
        assert probes.is_cuda and probes.dtype == torch.uint8 and probes.is_contiguous() and probes.shape[-1] == 32
        assert probe_desc.is_cuda and probe_desc.dtype == torch.uint8 and probe_desc.is_contiguous()
        assert probes.shape[0] == F.batch and probe_desc.shape[:2] == probes.shape[:2]
        assert nprobes.is_cuda and nprobes.dtype == torch.int32
        dev = F.n.device
        pcap = int(probes.shape[1])
        best_idx = torch.empty((F.batch, pcap), dtype=torch.int32, device=dev)
        best_dist = torch.empty((F.batch, pcap), dtype=torch.int32, device=dev)
        self._begin()
        _lib.check(self._L.orbhip_best_in_radius_device(
            self._h, probes.data_ptr(), probe_desc.data_ptr(), nprobes.data_ptr(), pcap, F.kp.data_ptr(), F.desc.data_ptr(),
            F.cell_start.data_ptr(), F.cell_items.data_ptr(), F.batch, F.cap, F.bounds, best_idx.data_ptr(), best_dist.data_ptr()))
        self._end()
        return best_idx, best_dist

    def best_in_radius_host(self, probes, probe_desc, kp2, desc2, bounds):
        """One keyframe from host arrays (what the forwarding ORBmatcher::Fuse / SearchBySim3 call after projecting
        their map points).  Returns (best_idx, best_dist) per probe."""
        from .extractor import KP_DTYPE
        from .frame import PROBE_DTYPE
        probes = np.ascontiguousarray(probes, dtype=PROBE_DTYPE)
        probe_desc = np.ascontiguousarray(probe_desc, dtype=np.uint8).reshape(-1, 32)
        kp2 = np.ascontiguousarray(kp2, dtype=KP_DTYPE)
        desc2 = np.ascontiguousarray(desc2, dtype=np.uint8).reshape(-1, 32)
        bi = np.full(max(len(probes), 1), -1, np.int32)
        bd = np.full(max(len(probes), 1), INT_MAX, np.int32)
        _lib.check(self._L.orbhip_best_in_radius(self._h, _p(probes), _p(probe_desc), len(probes), _p(kp2), _p(desc2), len(kp2),
                                                 _lib.OrbFrameBounds(*map(int, bounds)), _p(bi), _p(bd)))
        return bi[:len(probes)].copy(), bd[:len(probes)].copy()

    def search_by_projection_host(self, probes, probe_desc, kp2, desc2, bounds, rule, taken=None, th_dist=TH_HIGH,
                                  check_orientation=None):
        """One target frame from host arrays (what a forwarding ORBmatcher::SearchByProjection does after
        projecting its map points).  Returns (nmatches, match_of_feature)."""
        from .extractor import KP_DTYPE
        from .frame import PROBE_DTYPE
        probes = np.ascontiguousarray(probes, dtype=PROBE_DTYPE)
        probe_desc = np.ascontiguousarray(probe_desc, dtype=np.uint8).reshape(-1, 32)
        kp2 = np.ascontiguousarray(kp2, dtype=KP_DTYPE)
        desc2 = np.ascontiguousarray(desc2, dtype=np.uint8).reshape(-1, 32)
        tk = None if taken is None else np.ascontiguousarray(taken, dtype=np.uint8)
        out = np.full(max(len(kp2), 1), -1, np.int32)
        nm = C.c_int32(0)
        ori = self.mbCheckOrientation if check_orientation is None else check_orientation
        _lib.check(self._L.orbhip_search_by_projection(
            self._h, _p(probes), _p(probe_desc), len(probes), _p(kp2), _p(desc2), len(kp2),
            _lib.OrbFrameBounds(*map(int, bounds)), None if tk is None else _p(tk), int(rule), self.mfNNratio, int(th_dist),
            int(ori), _p(out), C.byref(nm)))
        return nm.value, out[:len(kp2)].copy()

    def search_for_initialization_host(self, kp1, desc1, kp2, desc2, bounds, prev_matched, windowSize=10):
        """One frame pair from host arrays (what a forwarding ORBmatcher::SearchForInitialization does).
        Returns (nmatches, vnMatches12, updated vbPrevMatched)."""
        from .extractor import KP_DTYPE
        kp1 = np.ascontiguousarray(kp1, dtype=KP_DTYPE)
        kp2 = np.ascontiguousarray(kp2, dtype=KP_DTYPE)
        desc1 = np.ascontiguousarray(desc1, dtype=np.uint8).reshape(-1, 32)
        desc2 = np.ascontiguousarray(desc2, dtype=np.uint8).reshape(-1, 32)
        prev = np.array(prev_matched, dtype=np.float32).reshape(-1, 2).copy()
        m12 = np.full(max(len(kp1), 1), -1, np.int32)
        nm = C.c_int32(0)
        _lib.check(self._L.orbhip_search_for_initialization(
            self._h, _p(kp1), _p(desc1), len(kp1), _p(kp2), _p(desc2), len(kp2), _lib.OrbFrameBounds(*map(int, bounds)),
            _p(prev), int(windowSize), self.mfNNratio, int(self.mbCheckOrientation), _p(m12), C.byref(nm)))
        return nm.value, m12[:len(kp1)].copy(), prev

    def window_search_host(self, kp1, desc1, valid1, kp2, desc2, bounds, windowSize, minScaleLevel=-1,
                           maxScaleLevel=INT_MAX):
        """One frame pair from host arrays (what a forwarding ORBmatcher::WindowSearch does).
        Returns (nmatches, vnMatches21)."""
        from .extractor import KP_DTYPE
        kp1 = np.ascontiguousarray(kp1, dtype=KP_DTYPE)
        kp2 = np.ascontiguousarray(kp2, dtype=KP_DTYPE)
        desc1 = np.ascontiguousarray(desc1, dtype=np.uint8).reshape(-1, 32)
        desc2 = np.ascontiguousarray(desc2, dtype=np.uint8).reshape(-1, 32)
        valid1 = np.ascontiguousarray(valid1, dtype=np.uint8)
        m21 = np.full(max(len(kp2), 1), -1, np.int32)
        nm = C.c_int32(0)
        _lib.check(self._L.orbhip_window_search(
            self._h, _p(kp1), _p(desc1), len(kp1), _p(valid1), _p(kp2), _p(desc2), len(kp2),
            _lib.OrbFrameBounds(*map(int, bounds)), int(windowSize), int(minScaleLevel), int(maxScaleLevel),
            self.mfNNratio, int(self.mbCheckOrientation), _p(m21), C.byref(nm)))
        return nm.value, m21[:len(kp2)].copy()

    def SearchByBoW(self, F1, F2, valid1, valid2=None, kf_frame=False):
        """ORBmatcher::SearchByBoW for every pair of two FrameBatch objects on which ComputeBoW has run.
        kf_frame=False: SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (src/ORBmatcher.cc:726-861), valid1 / valid2
        uint8 CUDA tensors [batch, cap] (map point exists and is not bad).
        kf_frame=True: SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (:158-295), F1 the keyframe, F2 the frame.
        Returns (nmatches [batch], matches12 [batch, cap], matches21 [batch, cap]) int32 CUDA tensors."""
        import torch
        assert F1.batch == F2.batch and F1.cap == F2.cap and F1.node_cap == F2.node_cap
        dev = F1.n.device
        m12 = torch.empty((F1.batch, F1.cap), dtype=torch.int32, device=dev)
        m21 = torch.empty((F1.batch, F1.cap), dtype=torch.int32, device=dev)
        nm = torch.empty(F1.batch, dtype=torch.int32, device=dev)
        self._begin()
        _lib.check(self._L.orbhip_search_by_bow_device(
            self._h, F1.kp.data_ptr(), F1.desc.data_ptr(), F1.n.data_ptr(), valid1.data_ptr(), F1.fv_nodes.data_ptr(),
            F1.fv_start.data_ptr(), F1.fv_feat.data_ptr(), F1.fv_count.data_ptr(), F2.kp.data_ptr(), F2.desc.data_ptr(),
            F2.n.data_ptr(), None if (kf_frame or valid2 is None) else valid2.data_ptr(), F2.fv_nodes.data_ptr(),
            F2.fv_start.data_ptr(), F2.fv_feat.data_ptr(), F2.fv_count.data_ptr(), F1.batch, F1.cap, F1.node_cap,
            self.mfNNratio, int(self.mbCheckOrientation), int(bool(kf_frame)), m12.data_ptr(), m21.data_ptr(),
            nm.data_ptr()))
        self._end()
        return nm, m12, m21

    def SearchForTriangulation(self, F1, F2, free1, free2, F12, level_sigma2):
        """ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:863-1025) for every pair of two FrameBatch objects on
        which ComputeBoW has run.  free1 / free2: uint8 CUDA tensors [batch, cap], non-zero where the keyframe's feature
        holds no map point; F12: float32 CUDA tensor [batch, 3, 3]; level_sigma2: float32 CUDA tensor [nlevels]
        (KeyFrame::GetSigma2).  Returns (nmatches [batch], matches12 [batch, cap] = vMatches12, matches21)."""
        import torch
        assert F1.batch == F2.batch and F1.cap == F2.cap and F1.node_cap == F2.node_cap
        assert F12.is_cuda and F12.dtype == torch.float32 and F12.is_contiguous() and F12.numel() == 9 * F1.batch
        assert level_sigma2.is_cuda and level_sigma2.dtype == torch.float32 and level_sigma2.is_contiguous()
        for fr in (free1, free2):
            assert fr.is_cuda and fr.dtype == torch.uint8 and fr.is_contiguous() and tuple(fr.shape) == (F1.batch, F1.cap)
        dev = F1.n.device
        m12 = torch.empty((F1.batch, F1.cap), dtype=torch.int32, device=dev)
        m21 = torch.empty((F1.batch, F1.cap), dtype=torch.int32, device=dev)
        nm = torch.empty(F1.batch, dtype=torch.int32, device=dev)
        self._begin()
        _lib.check(self._L.orbhip_search_for_triangulation_device(
            self._h, F1.kp.data_ptr(), F1.desc.data_ptr(), F1.n.data_ptr(), free1.data_ptr(), F1.fv_nodes.data_ptr(),
            F1.fv_start.data_ptr(), F1.fv_feat.data_ptr(), F1.fv_count.data_ptr(), F2.kp.data_ptr(), F2.desc.data_ptr(),
            F2.n.data_ptr(), free2.data_ptr(), F2.fv_nodes.data_ptr(), F2.fv_start.data_ptr(), F2.fv_feat.data_ptr(),
            F2.fv_count.data_ptr(), F1.batch, F1.cap, F1.node_cap, F12.data_ptr(), level_sigma2.data_ptr(),
            int(level_sigma2.numel()), int(self.mbCheckOrientation), m12.data_ptr(), m21.data_ptr(), nm.data_ptr()))
        self._end()
        return nm, m12, m21

    def ComputeDistinctiveDescriptors(self, desc, start):
        """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:191-256) for many map points:
        desc [total, 32] u8 and start [npoints + 1] i32 CUDA tensors (observations of point p are rows
        start[p]..start[p+1]).  Returns (BestIdx, BestMedian) int32 CUDA tensors of length npoints."""
        import torch
        assert desc.is_cuda and start.is_cuda and start.dtype == torch.int32
        npts = int(start.numel()) - 1
        idx = torch.empty(max(npts, 1), dtype=torch.int32, device=start.device)
        med = torch.empty(max(npts, 1), dtype=torch.int32, device=start.device)
        self._begin()
        _lib.check(self._L.orbhip_distinctive_descriptors_device(self._h, desc.data_ptr(), start.data_ptr(), npts,
                                                                 idx.data_ptr(), med.data_ptr()))
        self._end()
        return idx[:npts], med[:npts]

    def valu_rate(self, mode):
        """Sustained lane-ops/s of v_bcnt (0), the xor+bcnt mix (1) or v_add (2) on this device."""
        return float(self._L.orbhip_valu_rate(self._h, mode))

    def synchronize(self):
        _lib.check(self._L.orbhip_matcher_synchronize(self._h))

    def last_elapsed_ms(self):
        return self._L.orbhip_matcher_last_elapsed_ms(self._h)
