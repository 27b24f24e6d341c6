"""Host-side mirror of the part of ORB_SLAM::Frame the windowed matchers use (include/Frame.h:37-38,
90-92, src/Frame.cc:78-79,110-124,201-278): the undistorted keypoints, the descriptors and the
64 x 48 feature grid, for a batch of independent frames resident in device memory.

torch tensors are plumbing only (device memory); every computation is a call into liborbhip.so.
"""
import ctypes as C

import numpy as np

from . import _lib

FRAME_GRID_ROWS = 48  # include/Frame.h:37
FRAME_GRID_COLS = 64  # include/Frame.h:38
GRID_CELLS = FRAME_GRID_ROWS * FRAME_GRID_COLS

AREA_QUERY_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("r", "<f4"), ("min_level", "<i4"), ("max_level", "<i4")])
# OrbProbe of include/orbhip.h: one projected map point of the SearchByProjection overloads
PROBE_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("r", "<f4"), ("min_level", "<i4"), ("max_level", "<i4"),
                        ("angle", "<f4"), ("valid", "<i4"), ("reserved", "<i4")])


def UndistortKeyPoints(matcher, kp, n, K, dist, out=None):
    """Frame::UndistortKeyPoints (src/Frame.cc:290-320) for a batch of frames.  kp: CUDA tensor of keypoint
    records [batch, cap] (28 B each), n: int32 [batch], K = (fx, fy, cx, cy), dist = (k1, k2, p1, p2).
    Returns mvKeysUn as a tensor shaped like kp (`out` may be kp itself)."""
    import torch
    assert kp.is_cuda and kp.is_contiguous() and n.is_cuda and n.dtype == torch.int32
    batch = int(n.shape[0])
    cap = kp.numel() * kp.element_size() // 28 // max(batch, 1)
    out = torch.empty_like(kp) if out is None else out
    cam = _lib.OrbCamera(*[float(v) for v in K], *[float(v) for v in dist])
    matcher._begin()
    _lib.check(_lib.lib().orbhip_undistort_keypoints_device(matcher._h, kp.data_ptr(), n.data_ptr(), batch, cap, cam,
                                                            out.data_ptr()))
    matcher._end()
    return out


def ComputeImageBounds(K, dist, cols, rows):
    """Frame::ComputeImageBounds (src/Frame.cc:322-350): [mnMinX, mnMinY, mnMaxX, mnMaxY]."""
    cam = _lib.OrbCamera(*[float(v) for v in K], *[float(v) for v in dist])
    b = _lib.OrbFrameBounds()
    _lib.check(_lib.lib().orbhip_image_bounds(cam, int(cols), int(rows), b))
    return [b.min_x, b.min_y, b.max_x, b.max_y]


class FrameBatch:
    """`batch` frames: keypoint records [batch, cap] (28 B each, as the extractor writes them),
    descriptors [batch, cap, 32] u8, counts [batch] i32 — torch CUDA tensors — plus the image bounds
    (mnMinX, mnMinY, mnMaxX, mnMaxY).  The grid is built on construction like the reference's Frame
    constructor does."""

    def __init__(self, matcher, kp, desc, n, bounds):
        import torch
        self._m = matcher
        self.kp, self.desc, self.n = kp, desc, n
        self.batch, self.cap = int(n.shape[0]), int(desc.shape[1])
        assert kp.is_cuda and desc.is_cuda and n.is_cuda and n.dtype == torch.int32
        assert kp.numel() * kp.element_size() == self.batch * self.cap * 28
        self.bounds = _lib.OrbFrameBounds(*[int(b) for b in bounds])
        self.cell_start = torch.zeros((self.batch, GRID_CELLS + 1), dtype=torch.int32, device=n.device)
        self.cell_items = torch.zeros((self.batch, self.cap), dtype=torch.int32, device=n.device)
        L = _lib.lib()
        matcher._begin()
        _lib.check(L.orbhip_frame_grid_device(matcher._h, kp.data_ptr(), n.data_ptr(), self.batch, self.cap, self.bounds,
                                              self.cell_start.data_ptr(), self.cell_items.data_ptr()))
        matcher._end()

    @classmethod
    def from_host(cls, matcher, kps, descs, bounds, cap=None, device="cuda:0"):
        """kps: list of structured keypoint arrays, descs: list of [n, 32] u8 arrays."""
        import torch
        batch = len(kps)
        cap = cap or max(1, max(len(k) for k in kps))
        kp = np.zeros((batch, cap, 7), np.float32)
        de = np.zeros((batch, cap, 32), np.uint8)
        n = np.zeros(batch, np.int32)
        for f, (k, d) in enumerate(zip(kps, descs)):
            n[f] = len(k)
            if len(k):
                kp[f, :len(k)] = np.ascontiguousarray(k).view(np.float32).reshape(len(k), 7)
                de[f, :len(k)] = d
        return cls(matcher, torch.from_numpy(kp).to(device), torch.from_numpy(de).to(device),
                   torch.from_numpy(n).to(device), bounds)

    @classmethod
    def from_images(cls, extractor, matcher, images, K=None, dist=None):
        """The image part of the Frame constructor (src/Frame.cc:50-128) for a batch of equally sized frames that
        are resident on the device: ORB extraction (:61), UndistortKeyPoints (:72), the image bounds
        (:75-88, ComputeImageBounds) and the assignment of the undistorted keypoints to the grid (:110-124), with
        no host round trip in between.  images: uint8 CUDA tensor [batch, rows, cols]; K = (fx, fy, cx, cy) and
        dist = (k1, k2, p1, p2) as in mK / mDistCoef (None: no distortion).  The extractor must have been created
        with max_batch >= 1 for this image size.  The result holds mvKeys in .kp_raw and mvKeysUn in .kp."""
        import torch
        assert images.is_cuda and images.dtype == torch.uint8 and images.dim() == 3 and images.is_contiguous()
        b, rows, cols = (int(v) for v in images.shape)
        cap = extractor.nfeatures
        dev = images.device
        kp = torch.zeros((b, cap, 7), dtype=torch.float32, device=dev)
        desc = torch.zeros((b, cap, 32), dtype=torch.uint8, device=dev)
        n = torch.zeros(b, dtype=torch.int32, device=dev)
        # with a matcher in its asynchronous form the whole chain is queued: streams wait for one another on the device
        queued = getattr(matcher, "_async", False)
        if queued:
            extractor.torch_stream().wait_stream(torch.cuda.current_stream(dev))
        else:
            torch.cuda.synchronize()
        extractor.extract_batch_device(images.data_ptr(), b, rows, cols, cols, rows * cols, kp.data_ptr(), desc.data_ptr(),
                                       cap, n.data_ptr())
        if queued:
            torch.cuda.current_stream(dev).wait_stream(extractor.torch_stream())
        else:
            extractor.synchronize()
        if K is None or dist is None or float(dist[0]) == 0.0:
            kp_un, bounds = kp, [0, 0, cols, rows]
        else:
            kp_un = UndistortKeyPoints(matcher, kp, n, K, dist)
            bounds = ComputeImageBounds(K, dist, cols, rows)
        fb = cls(matcher, kp_un, desc, n, bounds)
        fb.kp_raw = kp
        return fb

    def ComputeBoW(self, voc, levelsup=4, node_cap=None):
        """Frame::ComputeBoW / KeyFrame::ComputeBoW (src/Frame.cc:281-288, src/KeyFrame.cc:82-91) for every frame
        of the batch: the vocabulary descent of all descriptors and the FeatureVector (mFeatVec) on the device.
        Sets bow_word / bow_node / bow_weight [batch, cap] and the FeatureVector arrays fv_nodes [batch, node_cap],
        fv_start [batch, node_cap + 1], fv_feat [batch, cap], fv_count [batch]."""
        import torch
        dev = self.n.device
        b, cap = self.batch, self.cap
        self.node_cap = int(node_cap or cap)
        self.bow_word = torch.zeros((b, cap), dtype=torch.int32, device=dev)
        self.bow_node = torch.zeros((b, cap), dtype=torch.int32, device=dev)
        self.bow_weight = torch.zeros((b, cap), dtype=torch.float64, device=dev)
        self.fv_nodes = torch.zeros((b, self.node_cap), dtype=torch.int32, device=dev)
        self.fv_start = torch.zeros((b, self.node_cap + 1), dtype=torch.int32, device=dev)
        self.fv_feat = torch.zeros((b, cap), dtype=torch.int32, device=dev)
        self.fv_count = torch.zeros(b, dtype=torch.int32, device=dev)
        self._m._begin()
        # descriptors past a frame's count are transformed too (rows of zeros) and ignored by the FeatureVector
        voc.transform_device(self._m, self.desc.data_ptr(), b * cap, levelsup, self.bow_word.data_ptr(),
                             self.bow_node.data_ptr(), self.bow_weight.data_ptr())
        _lib.check(_lib.lib().orbhip_feature_vector_device(
            self._m._h, self.bow_node.data_ptr(), self.bow_weight.data_ptr(), self.n.data_ptr(), b, cap, self.node_cap,
            self.fv_nodes.data_ptr(), self.fv_start.data_ptr(), self.fv_feat.data_ptr(), self.fv_count.data_ptr()))
        self._m._end()

    def feature_vector(self, f):
        """mFeatVec of frame f on the host: (nodes, start, features)."""
        k = int(self.fv_count[f])
        st = self.fv_start[f, :k + 1].cpu().numpy()
        return self.fv_nodes[f, :k].cpu().numpy(), st, self.fv_feat[f, :st[-1] if k else 0].cpu().numpy()

    def grid(self, f):
        """(cell_start[3073], items) of frame f on the host; cell = ix*48 + iy."""
        cs = self.cell_start[f].cpu().numpy()
        return cs, self.cell_items[f, :cs[-1]].cpu().numpy()

    def GetFeaturesInArea(self, queries, out_cap=None):
        """Batched Frame::GetFeaturesInArea.  queries: [batch, nq] array of AREA_QUERY_DTYPE.
        Returns (starts [batch, nq+1], indices [batch, out_cap]) as host arrays."""
        import torch
        q = np.ascontiguousarray(queries, dtype=AREA_QUERY_DTYPE).reshape(self.batch, -1)
        nq = q.shape[1]
        out_cap = out_cap or max(1, nq * self.cap)
        dev = self.n.device
        d_q = torch.from_numpy(q.view(np.uint8).reshape(self.batch, -1)).to(dev)
        d_start = torch.zeros((self.batch, nq + 1), dtype=torch.int32, device=dev)
        d_out = torch.zeros((self.batch, out_cap), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        _lib.check(_lib.lib().orbhip_features_in_area_device(
            self._m._h, self.kp.data_ptr(), self.n.data_ptr(), self.batch, self.cap, self.bounds,
            self.cell_start.data_ptr(), self.cell_items.data_ptr(), d_q.data_ptr(), nq, d_start.data_ptr(),
            d_out.data_ptr(), out_cap))
        self._m.synchronize()
        return d_start.cpu().numpy(), d_out.cpu().numpy()
