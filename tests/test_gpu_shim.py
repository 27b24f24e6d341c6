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
    o = 8
    kp = np.frombuffer(blob[o:o + 28 * n], orb.KP_DTYPE)
    o += 28 * n
    desc = np.frombuffer(blob[o:o + 32 * n], np.uint8).reshape(n, 32)
    o += 32 * n
    rkp, rdesc = orbref.RefExtractor(1000, 1.2, 8, 1, 20)(img)
    assert n == len(rkp)
    assert np.array_equal(kp.view(np.uint32), rkp.view(np.uint32))
    assert np.array_equal(desc, rdesc)
    assert dist01 == orbref.hamming(rdesc[0], rdesc[1])

    # second frame + the forwarding SearchForInitialization / WindowSearch bodies (shim/ORBmatcher.cc)
    n2 = int(np.frombuffer(blob[o:o + 4], np.int32)[0])
    o += 4
    kp2 = np.frombuffer(blob[o:o + 28 * n2], orb.KP_DTYPE)
    o += 28 * n2
    desc2 = np.frombuffer(blob[o:o + 32 * n2], np.uint8).reshape(n2, 32)
    o += 32 * n2
    n_init = int(np.frombuffer(blob[o:o + 4], np.int32)[0])
    o += 4
    m12 = np.frombuffer(blob[o:o + 4 * n], np.int32)
    o += 4 * n
    prev = np.frombuffer(blob[o:o + 8 * n], np.float32).reshape(n, 2)
    o += 8 * n
    n_win = int(np.frombuffer(blob[o:o + 4], np.int32)[0])
    o += 4
    has2 = np.frombuffer(blob[o:o + 4 * n2], np.int32)
    img2 = np.roll(img, (3, 5), axis=(0, 1))
    rkp2, rdesc2 = orbref.RefExtractor(1000, 1.2, 8, 1, 20)(img2)
    assert np.array_equal(kp2.view(np.uint32), rkp2.view(np.uint32)) and np.array_equal(desc2, rdesc2)
    bounds = [0, 0, 640, 480]
    prev0 = np.stack([rkp["x"], rkp["y"]], 1)
    rn, r12, rprev = orbref.search_for_initialization(rkp, rdesc, rkp2, rdesc2, bounds, prev0, 100, np.float32(0.9), True)
    assert n_init == rn and rn > 20
    assert np.array_equal(m12, r12) and np.array_equal(prev.view(np.uint32), rprev.view(np.uint32))
    idx = np.arange(n)
    valid = ((idx % 7 != 0) & (idx % 11 != 0)).astype(np.uint8)
    rn, r21 = orbref.window_search(rkp, rdesc, valid, rkp2, rdesc2, bounds, 50, 0, 2**31 - 1, np.float32(0.9), True)
    assert n_win == rn and rn > 20
    assert np.array_equal(has2 != 0, r21 >= 0)
    o += 4 * n2

    # the forwarding SearchByProjection(Frame&, vector<MapPoint*>&, th) body
    n_proj = int(np.frombuffer(blob[o:o + 4], np.int32)[0])
    o += 4
    owner = np.frombuffer(blob[o:o + 4 * n2], np.int32)
    scale = np.ones(8, np.float32)
    for l in range(1, 8):
        scale[l] = scale[l - 1] * np.float32(1.2)
    taken = (np.arange(n2) % 9 == 0).astype(np.uint8)
    rn, rmatch, _ = orbref.sbp_local_map(
        rkp2, rdesc2, bounds, scale, taken, (idx % 5 != 0).astype(np.uint8), (idx % 13 == 0).astype(np.uint8),
        rkp["octave"], np.where(idx % 3 == 0, np.float32(0.9985), np.float32(0.99)).astype(np.float32),
        rkp["x"] + np.float32(5), rkp["y"] + np.float32(3), rdesc, 3.0, np.float32(0.8))
    assert n_proj == rn and rn > 50
    assert np.array_equal(owner, np.where(taken != 0, -2, rmatch))
