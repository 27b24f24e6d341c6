"""Asynchronous form of the frame chain: with a candidate reservation on the matcher
(orbhip_matcher_reserve_candidates) extraction, grid assignment, SearchForInitialization, WindowSearch and
SearchByProjection of a batch are queued without any host wait, and one synchronisation at the end returns the
same results as the blocking form (which the other tests compare with the oracle).  A reservation that is too
small fails loudly at that synchronisation and yields no matches."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _chain(orb, torch, ext, m, img1, img2, w, h):
    F1 = orb.FrameBatch.from_images(ext, m, img1)
    F2 = orb.FrameBatch.from_images(ext, m, img2)
    b, cap = F1.batch, F1.cap
    prev = F1.kp.view(torch.float32).view(b, cap, 7)[:, :, 0:2].contiguous()
    nm_i, m12 = m.SearchForInitialization(F1, F2, prev, 60)
    valid = torch.ones((b, cap), dtype=torch.uint8, device="cuda")
    nm_w, m21 = m.WindowSearch(F1, F2, 40, valid, 0)
    # one probe per keypoint of F1 at its own position, as the last-frame overload builds them for identity motion
    k1 = F1.kp.view(torch.float32).view(b, cap, 7)
    octv = F1.kp.view(torch.int32).view(b, cap, 7)[:, :, 5]
    pr = torch.zeros((b, cap, 8), dtype=torch.float32, device="cuda")
    pri = pr.view(torch.int32)
    pr[:, :, 0:2] = k1[:, :, 0:2]
    pr[:, :, 2] = 12.0
    pri[:, :, 3] = octv - 1
    pri[:, :, 4] = octv + 1
    pr[:, :, 5] = k1[:, :, 3]
    pri[:, :, 6] = 1
    nm_p, mo = m.SearchByProjection(F2, pr.view(torch.uint8).view(b, cap, 32), F1.desc, F1.n, orb.ORBmatcher.PROJ_BEST,
                                    th_dist=100)
    return F1, F2, (nm_i, m12, prev, nm_w, m21, nm_p, mo)


def test_queued_chain_equals_blocking_chain(orb, synth):
    import torch
    b, w, h, nf = 6, 320, 240, 600
    imgs = np.stack([synth.synth_image(w, h, 40 + i) for i in range(b)])
    img1 = torch.from_numpy(imgs).cuda()
    img2 = torch.from_numpy(np.roll(imgs, (2, 3), axis=(1, 2)).copy()).cuda()
    ext = orb.ORBextractor(nf, 1.2, 8, 1, 20, max_width=w, max_height=h, max_batch=b)
    blocking = orb.ORBmatcher(0.9, True)
    _, _, ref = _chain(orb, torch, ext, blocking, img1, img2, w, h)
    ref = [t.cpu().numpy() for t in ref]
    queued = orb.ORBmatcher(0.9, True)
    queued.reserve_candidates(4 << 20)
    for _ in range(2):  # the second round reuses the workspace of the first
        F1, F2, got = _chain(orb, torch, ext, queued, img1, img2, w, h)
        queued.synchronize()
        ext.synchronize()
        torch.cuda.synchronize()
        for a, g in zip(ref, got):
            assert np.array_equal(a, g.cpu().numpy())
    assert int(ref[0].sum()) > 100 and int(ref[3].sum()) > 100 and int(ref[5].sum()) > 100


def test_reservation_too_small_fails_at_synchronize(orb, synth):
    import torch
    b, w, h, nf = 2, 320, 240, 600
    imgs = np.stack([synth.synth_image(w, h, 50 + i) for i in range(b)])
    img = torch.from_numpy(imgs).cuda()
    ext = orb.ORBextractor(nf, 1.2, 8, 1, 20, max_width=w, max_height=h, max_batch=b)
    m = orb.ORBmatcher(0.9, True)
    m.reserve_candidates(64)
    F1 = orb.FrameBatch.from_images(ext, m, img)
    # SearchForInitialization keeps candidate lists (WindowSearch and the projection matchers store none)
    prev = F1.kp.view(torch.float32).view(b, F1.cap, 7)[:, :, 0:2].contiguous()
    nm, m12 = m.SearchForInitialization(F1, F1, prev.clone(), 40)
    with pytest.raises(orb.OrbhipError) as e:
        m.synchronize()
    assert "reservation" in str(e.value) and e.value.code == orb._lib.ECAP
    torch.cuda.synchronize()
    assert int(nm.sum()) == 0 and bool((m12 == -1).all())
    # the flag is cleared by the failed synchronisation; with room the same call works, and 0 returns to blocking
    m.reserve_candidates(4 << 20)
    nm, m12 = m.SearchForInitialization(F1, F1, prev.clone(), 40)
    m.synchronize()
    torch.cuda.synchronize()
    assert int(nm.sum()) > 100
    m.reserve_candidates(0)
    nm2, m12b = m.SearchForInitialization(F1, F1, prev.clone(), 40)
    assert torch.equal(m12, m12b)
    # the projection matchers are indifferent to the reservation
    m.reserve_candidates(64)
    pr = torch.zeros((b, F1.cap, 8), dtype=torch.float32, device="cuda")
    pr[:, :, 0:2] = prev
    pr[:, :, 2] = 20.0
    pr.view(torch.int32)[:, :, 3:5] = -1
    pr.view(torch.int32)[:, :, 6] = 1
    nmp, _ = m.SearchByProjection(F1, pr.view(torch.uint8).view(b, F1.cap, 32), F1.desc, F1.n, orb.ORBmatcher.PROJ_BEST, th_dist=100)
    m.synchronize()
    torch.cuda.synchronize()
    assert int(nmp.sum()) > 100
