"""Loader and ctypes prototypes of liborbhip.so (the C ABI of include/orbhip.h).

There is no CPU fallback: if the shared library is missing the import of the product classes
fails with an ImportError that says how to build it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORBHIP_LIBRARY names another build of the same library (developer builds with instrumentation)
LIB_PATH = os.environ.get("ORBHIP_LIBRARY") or os.path.join(_HERE, "liborbhip.so")

OK, EINVAL, ENOMEM, EHIP, ECAP, EGEOM, ENODEV, EUNSUPPORTED = 0, -1, -2, -3, -4, -5, -6, -7

_lib = None


class OrbhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("orbhip error %d: %s" % (code, msg))
        self.code = code


def _preload_torch_hip_runtime():
    """PyTorch's ROCm wheel bundles its own libamdhip64.so with the same SONAME as the system one.
    Two HIP runtimes in one process do not both see the GPU, so when torch is installed its copy
    is loaded first (without importing torch) and liborbhip.so binds to it; torch then reuses it."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass
    # the copy of RCCL torch would load: named here so that liborbhip.so binds to the same one
    rccl = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so")
    if os.path.exists(rccl):
        os.environ.setdefault("ORBHIP_RCCL_LIB", rccl)


class OrbCamera(C.Structure):
    """mK (fx, fy, cx, cy) and mDistCoef (k1, k2, p1, p2) of a frame, both CV_32F in the reference."""
    _fields_ = [(k, C.c_float) for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2")]


class OrbFrameBounds(C.Structure):
    """Frame::mnMinX, mnMinY, mnMaxX, mnMaxY (src/Frame.cc:322-350)."""
    _fields_ = [("min_x", C.c_int32), ("min_y", C.c_int32), ("max_x", C.c_int32), ("max_y", C.c_int32)]


class OrbProjection(C.Structure):
    """A pose (rotation, translation, camera centre; CV_32F) with the intrinsics: what the projection matchers use of
    a Frame / KeyFrame (src/Frame.cc:131-136, src/KeyFrame.cc:102-109)."""
    _fields_ = [("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Ow", C.c_float * 3), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float)]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `make -C %s` (hipcc, gfx950); there is no CPU fallback"
                          % (LIB_PATH, os.path.join(_HERE, "csrc")))
    _preload_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, i32, f32, pd, sz = C.c_void_p, C.c_int, C.c_float, C.c_ssize_t, C.c_size_t
    L.orbhip_extractor_create.argtypes = [C.POINTER(vp), i32, f32, i32, i32, i32, i32, i32, i32, i32]
    L.orbhip_extractor_destroy.argtypes = [vp]
    L.orbhip_extractor_destroy.restype = None
    L.orbhip_levels.argtypes = [vp]
    L.orbhip_scale_factor.argtypes = [vp]
    L.orbhip_scale_factor.restype = f32
    L.orbhip_table.argtypes = [vp, i32, vp]
    L.orbhip_extract.argtypes = [vp, vp, i32, i32, pd, vp, vp, i32, vp]
    L.orbhip_extract_batch.argtypes = [vp, vp, i32, i32, i32, pd, pd, vp, vp, i32, vp]
    L.orbhip_extract_batch_device.argtypes = [vp, vp, i32, i32, i32, pd, pd, vp, vp, i32, vp]
    L.orbhip_extractor_synchronize.argtypes = [vp]
    L.orbhip_extractor_stream.argtypes = [vp]
    L.orbhip_extractor_stream.restype = vp
    L.orbhip_debug_get.argtypes = [vp, i32, i32, i32, vp, sz]
    L.orbhip_level_geometry.argtypes = [vp, i32, vp]
    L.orbhip_last_elapsed_ms.argtypes = [vp]
    L.orbhip_last_elapsed_ms.restype = f32
    L.orbhip_hamming_distance.argtypes = [vp, vp]
    L.orbhip_matcher_create.argtypes = [C.POINTER(vp), i32]
    L.orbhip_matcher_destroy.argtypes = [vp]
    L.orbhip_matcher_destroy.restype = None
    L.orbhip_matcher_synchronize.argtypes = [vp]
    L.orbhip_matcher_stream.argtypes = [vp]
    L.orbhip_matcher_stream.restype = vp
    L.orbhip_matcher_reserve_candidates.argtypes = [vp, C.c_longlong]
    L.orbhip_matcher_set_scan_kernel.argtypes = [vp, C.c_int]
    L.orbhip_knn2_device.argtypes = [vp, vp, i32, pd, vp, i32, pd, i32, i32, vp, vp, vp]
    L.orbhip_knn2_device_noalloc.argtypes = [vp, vp, i32, pd, vp, i32, pd, i32, i32, vp, vp, vp]
    L.orbhip_knn2_scratch_bytes.argtypes = []
    L.orbhip_knn2_partial_bytes.argtypes = [vp, i32, i32, i32]
    L.orbhip_knn2_partial_bytes.restype = C.c_longlong
    L.orbhip_knn2.argtypes = [vp, vp, i32, vp, i32, i32, vp, vp, vp]
    L.orbhip_knn2_merge_device.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    L.orbhip_knn2_merge_host.argtypes = [vp, i32, i32, vp, vp, vp]
    L.orbhip_knn2_gather_layout.argtypes = [i32, i32, i32, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    L.orbhip_comm_unique_id.argtypes = [vp, sz]
    L.orbhip_comm_create.argtypes = [C.POINTER(vp), vp, sz, i32, i32, i32]
    L.orbhip_comm_adopt.argtypes = [C.POINTER(vp), vp, i32, i32, i32]
    L.orbhip_comm_destroy.argtypes = [vp]
    L.orbhip_comm_destroy.restype = None
    L.orbhip_comm_rank.argtypes = [vp]
    L.orbhip_comm_world.argtypes = [vp]
    L.orbhip_knn2_sharded.argtypes = [vp, vp, vp, i32, vp, i32, i32, vp, vp, vp]
    L.orbhip_ratio_test_device.argtypes = [vp, vp, vp, i32, f32, i32, i32, vp]
    L.orbhip_matcher_last_elapsed_ms.argtypes = [vp]
    L.orbhip_matcher_last_elapsed_ms.restype = f32
    L.orbhip_valu_rate.argtypes = [vp, i32]
    L.orbhip_valu_rate.restype = C.c_double
    L.orbhip_extractor_set_profiling.argtypes = [vp, i32]
    L.orbhip_extractor_set_gaussian_profile.argtypes = [vp, i32]
    L.orbhip_extractor_stage_times.argtypes = [vp, vp, vp]
    L.orbhip_selftest.argtypes = [i32, vp, i32, i32, vp]
    L.orbhip_frame_grid_device.argtypes = [vp, vp, vp, i32, i32, OrbFrameBounds, vp, vp]
    L.orbhip_features_in_area_device.argtypes = [vp, vp, vp, i32, i32, OrbFrameBounds, vp, vp, vp, i32, vp, vp, i32]
    L.orbhip_search_for_initialization_device.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, OrbFrameBounds,
                                                          vp, i32, f32, i32, vp, vp]
    L.orbhip_search_for_initialization.argtypes = [vp, vp, vp, i32, vp, vp, i32, OrbFrameBounds, vp, i32, f32, i32, vp, vp]
    L.orbhip_window_search.argtypes = [vp, vp, vp, i32, vp, vp, vp, i32, OrbFrameBounds, i32, i32, i32, f32, i32, vp, vp]
    L.orbhip_vocabulary_create.argtypes = [C.POINTER(vp), i32, i32, vp, vp, vp, i32]
    L.orbhip_vocabulary_destroy.argtypes = [vp]
    L.orbhip_vocabulary_destroy.restype = None
    L.orbhip_vocabulary_words.argtypes = [vp]
    L.orbhip_feature_vector_device.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]
    L.orbhip_search_by_bow_device.argtypes = [vp] + [vp] * 16 + [i32, i32, i32, f32, i32, i32, vp, vp, vp]
    L.orbhip_bow_transform_device.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp]
    L.orbhip_distinctive_descriptors_device.argtypes = [vp, vp, vp, i32, vp, vp]
    L.orbhip_window_search_device.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, OrbFrameBounds,
                                              i32, i32, i32, f32, i32, vp, vp]
    L.orbhip_search_by_projection_device.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, OrbFrameBounds,
                                                     vp, i32, f32, i32, i32, vp, vp]
    L.orbhip_search_by_projection.argtypes = [vp, vp, vp, i32, vp, vp, i32, OrbFrameBounds, vp, i32, f32, i32, i32, vp, vp]
    L.orbhip_search_for_triangulation_device.argtypes = [vp] + [vp] * 16 + [i32, i32, i32, vp, vp, i32, i32, vp, vp, vp]
    L.orbhip_search_for_triangulation.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, i32,
                                                  vp, vp, i32, i32, vp, vp]
    L.orbhip_projection_from_pose.argtypes = [vp, f32, f32, f32, f32, C.POINTER(OrbProjection)]
    L.orbhip_projection_from_sim3.argtypes = [vp, f32, f32, f32, f32, C.POINTER(OrbProjection)]
    L.orbhip_build_probes_device.argtypes = [vp, i32, vp, OrbFrameBounds, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, vp]
    L.orbhip_build_probes_sim3_device.argtypes = [vp, vp, vp, OrbFrameBounds, vp, i32, vp, vp, vp, vp, vp, i32, i32, f32, vp]
    L.orbhip_projection_sim3_pair.argtypes = [f32, vp, vp, f32, f32, f32, f32, C.POINTER(OrbProjection), C.POINTER(OrbProjection)]
    L.orbhip_probes_from_tracked.argtypes = [i32, vp, vp, vp, vp, f32, vp, i32, vp]
    L.orbhip_best_in_radius_device.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, OrbFrameBounds, vp, vp]
    L.orbhip_best_in_radius.argtypes = [vp, vp, vp, i32, vp, vp, i32, OrbFrameBounds, vp, vp]
    L.orbhip_undistort_keypoints_device.argtypes = [vp, vp, vp, i32, i32, OrbCamera, vp]
    L.orbhip_image_bounds.argtypes = [OrbCamera, i32, i32, C.POINTER(OrbFrameBounds)]
    L.orbhip_device_count.argtypes = []
    L.orbhip_version.restype = C.c_char_p
    L.orbhip_last_error.restype = C.c_char_p
    _lib = L
    return L


def check(rc):
    if rc < 0:
        raise OrbhipError(rc, lib().orbhip_last_error().decode("utf-8", "replace"))
    return rc
