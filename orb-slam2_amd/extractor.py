"""Host-side mirror of ORB_SLAM::ORBextractor (include/ORBextractor.h:32-77 of the reference) over
the HIP library: same constructor arguments, operator() -> __call__, GetLevels, GetScaleFactor.
"""
import ctypes as C

import numpy as np

from . import _lib

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class ORBextractor:
    HARRIS_SCORE, FAST_SCORE = 0, 1  # include/ORBextractor.h:36

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, scoreType=1, fastTh=20,
                 device=0, max_width=640, max_height=480, max_batch=1):
        self._h = C.c_void_p()
        self._L = _lib.lib()
        _lib.check(self._L.orbhip_extractor_create(C.byref(self._h), nfeatures, scaleFactor, nlevels, scoreType,
                                                   fastTh, device, max_width, max_height, max_batch))
        self.nfeatures = nfeatures
        self.nlevels = nlevels
        self.device = device

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.orbhip_extractor_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def GetLevels(self):
        return self._L.orbhip_levels(self._h)

    def GetScaleFactor(self):
        return self._L.orbhip_scale_factor(self._h)

    def table(self, which):
        out = np.zeros(64, dtype=np.float32 if which in (1, 2) else np.int32)
        n = _lib.check(self._L.orbhip_table(self._h, which, _p(out)))
        return out[:n].copy()

    def __call__(self, image, mask=None):
        """operator()(image, mask, keypoints, descriptors): returns (keypoints, descriptors).
        The mask has no effect on the reference's results (src/ORBextractor.cc:601-603) and is ignored.
        An empty image returns (None, None): the reference leaves its outputs untouched (:721-722)."""
        image = np.asarray(image)
        if image.size == 0:
            return None, None
        if image.dtype != np.uint8 or image.ndim != 2:
            raise AssertionError("image.type() == CV_8UC1")  # src/ORBextractor.cc:725
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        rows, cols = image.shape
        cap = self.nfeatures
        kp = np.zeros(cap, dtype=KP_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        n = C.c_int(0)
        _lib.check(self._L.orbhip_extract(self._h, _p(image), rows, cols, image.strides[0], _p(kp), _p(desc), cap,
                                          C.byref(n)))
        return kp[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, images):
        """images: u8 array [B, H, W] in host memory -> list of (keypoints, descriptors) per frame."""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        assert images.ndim == 3
        b, rows, cols = images.shape
        cap = self.nfeatures
        kp = np.zeros((b, cap), dtype=KP_DTYPE)
        desc = np.zeros((b, cap, 32), dtype=np.uint8)
        n = np.zeros(b, dtype=np.int32)
        _lib.check(self._L.orbhip_extract_batch(self._h, _p(images), b, rows, cols, images.strides[1], images.strides[0],
                                                _p(kp), _p(desc), cap, _p(n)))
        return [(kp[f, :n[f]].copy(), desc[f, :n[f]].copy()) for f in range(b)]

    def extract_batch_device(self, d_imgs, batch, rows, cols, row_stride, frame_stride, d_kp, d_desc, cap, d_n):
        """All arguments are raw device addresses (ints); asynchronous on the handle's stream."""
        _lib.check(self._L.orbhip_extract_batch_device(self._h, d_imgs, batch, rows, cols, row_stride, frame_stride,
                                                       d_kp, d_desc, cap, d_n))

    GAUSS_SCALAR, GAUSS_SSE2 = 0, 1

    def set_gaussian_profile(self, profile):
        """Column-pass rounding of the 7x7 Gaussian: GAUSS_SCALAR (default, the oracle's profile) or GAUSS_SSE2
        (ties to even in the first w & ~3 columns, as an SSE2 build of OpenCV 2.4 rounds them)."""
        _lib.check(self._L.orbhip_extractor_set_gaussian_profile(self._h, int(profile)))

    def torch_stream(self):
        """The extractor's stream as a torch stream (for device-side ordering against torch's own work)."""
        import torch
        if getattr(self, "_tstream", None) is None:
            self._tstream = torch.cuda.ExternalStream(self._L.orbhip_extractor_stream(self._h), device="cuda:%d" % self.device)
        return self._tstream

    def synchronize(self):
        _lib.check(self._L.orbhip_extractor_synchronize(self._h))

    def last_elapsed_ms(self):
        return self._L.orbhip_last_elapsed_ms(self._h)

    STAGES = ("level0", "resize", "fast_measure", "nms_compact", "select", "orient", "blur", "describe")

    def set_profiling(self, enable=True):
        _lib.check(self._L.orbhip_extractor_set_profiling(self._h, int(enable)))

    def stage_times(self):
        """{stage: (total ms, launches)} of the last synchronised call (profiling must be on)."""
        ms = np.zeros(8, np.float32)
        n = np.zeros(8, np.int32)
        _lib.check(self._L.orbhip_extractor_stage_times(self._h, _p(ms), _p(n)))
        return {s: (float(ms[i]), int(n[i])) for i, s in enumerate(self.STAGES)}

    # ---- stage taps (test instrumentation) ----
    def level_geometry(self, level):
        out = np.zeros(8, dtype=np.int32)
        _lib.check(self._L.orbhip_level_geometry(self._h, level, _p(out)))
        keys = ("w", "h", "cols", "rows", "cellW", "cellH", "nfeaturesCell", "quota")
        return dict(zip(keys, (int(v) for v in out)))

    def tap_plane(self, stage, frame, level):
        g = self.level_geometry(level)
        out = np.zeros((g["h"] + 32, g["w"] + 32), dtype=np.uint8)
        _lib.check(self._L.orbhip_debug_get(self._h, stage, frame, level, _p(out), out.nbytes))
        return out

    def tap_cells(self, stage, frame, level):
        g = self.level_geometry(level)
        out = np.zeros(g["cols"] * g["rows"], dtype=np.int32)
        _lib.check(self._L.orbhip_debug_get(self._h, stage, frame, level, _p(out), out.nbytes))
        return out

    def tap_level_keypoints(self, frame, level):
        g = self.level_geometry(level)
        out = np.zeros((g["quota"] + 1, 3), dtype=np.int32)
        n = _lib.check(self._L.orbhip_debug_get(self._h, 5, frame, level, _p(out), out.nbytes))
        return out[:n].copy()

    def tap_candidates(self, frame, level):
        g = self.level_geometry(level)
        out = np.zeros((g["w"] * g["h"] // 4 + 64, 3), dtype=np.int32)
        n = _lib.check(self._L.orbhip_debug_get(self._h, 6, frame, level, _p(out), out.nbytes))
        return out[:n].copy()
