"""The C-ABI library loads without a GPU, exports every function include/orbhip.h declares and
fails loudly (no CPU fallback) when asked to compute without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "orbhip.h")


def declared_functions():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(orbhip_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_boundary():
    names = declared_functions()
    for must in ("orbhip_extractor_create", "orbhip_extract", "orbhip_extract_batch", "orbhip_extract_batch_device",
                 "orbhip_levels", "orbhip_scale_factor", "orbhip_hamming_distance", "orbhip_knn2", "orbhip_knn2_device",
                 "orbhip_knn2_merge_device", "orbhip_ratio_test_device"):
        assert must in names


def test_library_exports_every_declared_symbol(orb):
    L = orb._lib.lib()
    for name in declared_functions():
        assert getattr(L, name) is not None, name


def test_struct_layout(orb):
    assert orb.KP_DTYPE.itemsize == 28
    assert [orb.KP_DTYPE.fields[f][1] for f in ("x", "y", "size", "angle", "response", "octave", "class_id")] == \
        [0, 4, 8, 12, 16, 20, 24]


def test_descriptor_distance_runs_on_host(orb):
    a = np.arange(32, dtype=np.uint8)
    b = a[::-1].copy()
    assert orb.ORBmatcher.DescriptorDistance(a, b) == int(np.unpackbits(a ^ b).sum())
    assert orb.ORBmatcher.TH_LOW == 20 and orb.ORBmatcher.TH_HIGH == 100 and orb.ORBmatcher.HISTO_LENGTH == 30


def test_no_cpu_fallback_without_gpu(orb):
    if orb._lib.lib().orbhip_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(orb.OrbhipError) as ei:
        orb.ORBextractor(1000, 1.2, 8, 1, 20)
    assert ei.value.code == orb._lib.ENODEV
    with pytest.raises(orb.OrbhipError):
        orb.ORBmatcher()


def test_bad_arguments_are_rejected_before_any_device_work(orb):
    L = orb._lib.lib()
    h = C.c_void_p()
    assert L.orbhip_extractor_create(C.byref(h), 0, 1.2, 8, 1, 20, 0, 640, 480, 1) == orb._lib.EINVAL
    assert L.orbhip_extractor_create(C.byref(h), 1000, 1.0, 8, 1, 20, 0, 640, 480, 1) == orb._lib.EINVAL
    assert L.orbhip_extractor_create(C.byref(h), 1000, 1.2, 99, 1, 20, 0, 640, 480, 1) == orb._lib.EINVAL
    assert L.orbhip_extractor_create(None, 1000, 1.2, 8, 1, 20, 0, 640, 480, 1) == orb._lib.EINVAL
    assert b"" != L.orbhip_last_error()
    assert L.orbhip_levels(None) < 0


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(HERE, "..", "orb-slam2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "orbref" not in txt and "oracle/" not in txt, f
