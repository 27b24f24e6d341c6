"""The C++ drop-in classes (orb-slam2_amd/shim): ORB_SLAM::ORBextractor::operator() called the way
src/Frame.cc:61 of the reference calls it, compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "..", "orb-slam2_amd", "shim", "shim_test")


def test_cpp_shim_matches_oracle(orb, orbref, synth, tmp_path):
    assert os.path.exists(EXE), "build it with make -C orb-slam2_amd/shim (done by __graft_entry__.build)"
    img = synth.synth_image(640, 480, 11)
    raw = tmp_path / "in.raw"
    out = tmp_path / "out.bin"
    raw.write_bytes(img.tobytes())
    r = subprocess.run([EXE, str(raw), "480", "640", "1000", str(out)], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    blob = out.read_bytes()
    n, dist01 = np.frombuffer(blob[:8], np.int32)
    kp = np.frombuffer(blob[8:8 + 28 * n], orb.KP_DTYPE)
    desc = np.frombuffer(blob[8 + 28 * n:], np.uint8).reshape(n, 32)
    rkp, rdesc = orbref.RefExtractor(1000, 1.2, 8, 1, 20)(img)
    assert n == len(rkp)
    assert np.array_equal(kp.view(np.uint32), rkp.view(np.uint32))
    assert np.array_equal(desc, rdesc)
    assert dist01 == orbref.hamming(rdesc[0], rdesc[1])
