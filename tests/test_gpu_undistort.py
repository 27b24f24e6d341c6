"""Frame::UndistortKeyPoints on the device (src/Frame.cc:290-320) against the oracle's restatement of
cv::undistortPoints, through the C ABI.  Bar: bit-exact (double arithmetic without contraction on both sides;
add, multiply and divide are correctly rounded on host and device)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CAMS = [((517.3, 516.5, 318.6, 255.3), (0.2624, -0.9531, -0.0054, 0.0026)),
        ((458.654, 457.296, 367.215, 248.375), (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)),
        ((700.0, 710.0, 640.0, 360.0), (-0.05, 0.0, 0.0, 0.0)),
        ((535.4, 539.2, 320.1, 247.6), (0.0, 0.2, 0.01, 0.01))]  # k1 == 0: plain copy


@pytest.mark.parametrize("cam", CAMS)
def test_undistort_keypoints(orb, orbref, cam):
    import torch
    K, D = cam
    rng = np.random.default_rng(17)
    sizes = [0, 1, 255, 256, 257, 3000]
    cap = max(sizes)
    kp = np.zeros((len(sizes), cap), orbref.KP_DTYPE)
    for f, n in enumerate(sizes):
        kp["x"][f, :n] = rng.uniform(-20, 1300, n).astype(np.float32)
        kp["y"][f, :n] = rng.uniform(-20, 740, n).astype(np.float32)
        kp["size"][f, :n], kp["angle"][f, :n] = 31.0, rng.uniform(0, 360, n)
        kp["response"][f, :n], kp["octave"][f, :n], kp["class_id"][f, :n] = rng.integers(1, 200, n), rng.integers(0, 8, n), -1
    m = orb.ORBmatcher(0.9, True)
    d_kp = torch.from_numpy(kp.view(np.float32).reshape(len(sizes), cap, 7).copy()).cuda()
    d_n = torch.tensor(sizes, dtype=torch.int32, device="cuda")
    sentinel = torch.full_like(d_kp, 7.0)
    out = orb.UndistortKeyPoints(m, d_kp, d_n, K, D, out=sentinel).cpu().numpy()
    moved = 0
    for f, n in enumerate(sizes):
        ref = orbref.undistort_keypoints(kp[f, :n], K, D)
        got = np.ascontiguousarray(out[f, :n]).view(orbref.KP_DTYPE).reshape(n)
        assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)), "frame %d" % f
        assert (out[f, n:] == 7.0).all()  # records past the count are not written
        moved += int((ref["x"] != kp["x"][f, :n]).sum())
    assert (moved > 3000) == (D[0] != 0.0)
    # in place
    inpl = orb.UndistortKeyPoints(m, d_kp.clone(), d_n, K, D)
    d2 = d_kp.clone()
    orb.UndistortKeyPoints(m, d2, d_n, K, D, out=d2)
    assert torch.equal(inpl[5, :3000].view(torch.int32), d2[5, :3000].view(torch.int32))  # class_id -1 is a NaN as float


def test_frame_constructor_path(orb, orbref, synth):
    """Extraction -> UndistortKeyPoints -> image bounds -> grid, all on the device (the image part of
    Frame::Frame, src/Frame.cc:50-128), against the same chain of oracle calls."""
    import torch
    w, h, nf = 640, 480, 800
    K, D = CAMS[0]
    imgs = synth.synth_batch(w, h, 3, first_idx=60)
    ext = orb.ORBextractor(nf, 1.2, 8, 1, 20, device=0, max_width=w, max_height=h, max_batch=3)
    m = orb.ORBmatcher(0.9, True)
    fb = orb.FrameBatch.from_images(ext, m, torch.from_numpy(imgs).cuda(), K, D)
    ref = orbref.RefExtractor(nf, 1.2, 8, 1, 20)
    rb = orbref.image_bounds(K, D, w, h)
    assert [fb.bounds.min_x, fb.bounds.min_y, fb.bounds.max_x, fb.bounds.max_y] == rb
    for f in range(3):
        rkp, rdesc = ref(imgs[f])
        run = orbref.undistort_keypoints(rkp, K, D)
        n = int(fb.n[f])
        assert n == len(rkp)
        got = np.ascontiguousarray(fb.kp[f, :n].cpu().numpy()).view(orbref.KP_DTYPE).reshape(n)
        assert np.array_equal(got.view(np.uint32), run.view(np.uint32))
        assert np.array_equal(fb.desc[f, :n].cpu().numpy(), rdesc)
        cs, ci = orbref.frame_grid(run, rb)
        ds, di = fb.grid(f)
        assert np.array_equal(cs, ds) and np.array_equal(ci, di)
    # no distortion: mvKeysUn is mvKeys and the bounds are the image
    fb0 = orb.FrameBatch.from_images(ext, m, torch.from_numpy(imgs).cuda())
    assert fb0.kp is fb0.kp_raw and [fb0.bounds.max_x, fb0.bounds.max_y] == [w, h]
