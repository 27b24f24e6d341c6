"""Device arithmetic helpers against the oracle / host libraries: fastAtan2, the glibc-exact
cosf/sinf, the libstdc++-exact nth_element and the FAST corner measure."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_fast_atan2_bit_equal(orb, orbref):
    rng = np.random.default_rng(0)
    n = 200000
    yx = rng.integers(-2900000, 2900001, size=(n, 2)).astype(np.float32)
    yx[:64] = [[0, 0], [0, 1], [1, 0], [-1, 0], [0, -1], [1, 1], [-1, -1], [5, -5]] * 8
    out = np.zeros(n, np.float32)
    orb._lib.check(orb._lib.lib().orbhip_selftest(0, _p(yx), n, 0, _p(out)))
    L = orbref.lib()
    ref = np.array([L.orbref_fast_atan2(float(y), float(x)) for y, x in yx[:20000]], np.float32)
    assert np.array_equal(out[:20000].view(np.uint32), ref.view(np.uint32))
    true = np.degrees(np.arctan2(yx[:, 0].astype(np.float64), yx[:, 1].astype(np.float64))) % 360.0
    nz = (yx[:, 0] != 0) | (yx[:, 1] != 0)
    err = np.abs(((out.astype(np.float64) - true + 180.0) % 360.0) - 180.0)[nz]
    assert err.max() < 0.3


def test_sincos_bit_equal_to_host_libm(orb, orbref):
    rng = np.random.default_rng(1)
    ang = np.concatenate([rng.uniform(0, 360, 300000).astype(np.float32) * np.float32(np.pi / 180),
                          np.array([0, 1e-6, 2 ** -12, 0.7499999, 0.75, 0.7853982, 1.5707964, 3.1415927, 6.2831855], np.float32)])
    n = len(ang)
    out = np.zeros((n, 2), np.float32)
    orb._lib.check(orb._lib.lib().orbhip_selftest(1, _p(ang), n, 0, _p(out)))
    L = orbref.lib()
    sub = np.concatenate([np.arange(0, 30000), np.arange(n - 9, n)])
    rc = np.array([L.orbref_cosf(float(a)) for a in ang[sub]], np.float32)
    rs = np.array([L.orbref_sinf(float(a)) for a in ang[sub]], np.float32)
    assert np.array_equal(out[sub, 0].view(np.uint32), rc.view(np.uint32))
    assert np.array_equal(out[sub, 1].view(np.uint32), rs.view(np.uint32))
    # the rest against numpy's float32 routines is only a sanity bound (not the same libm entry)
    assert np.abs(out[:, 0].astype(np.float64) - np.cos(ang.astype(np.float64))).max() < 1e-6


@pytest.mark.parametrize("n,mode", [(4, 0), (17, 0), (64, 1), (100, 0), (480, 0), (600, 1), (3000, 0), (5000, 2)])
def test_device_nth_element_matches_libstdcxx(orb, orbref, n, mode):
    rng = np.random.default_rng(n + mode)
    if mode == 0:
        score = rng.integers(7, 12, n)
    elif mode == 1:
        score = rng.integers(0, 255, n)
    else:
        score = np.where(rng.integers(0, 8, n) == 0, rng.integers(0, 255, n), 20)
    keys = (score.astype(np.uint32) << 24) | np.arange(n, dtype=np.uint32)
    L = orbref.lib()
    for k in sorted({1, n // 7 + 1, n // 2, n - 1}):
        if not 0 < k < n:
            continue
        out = np.zeros(n, np.uint32)
        orb._lib.check(orb._lib.lib().orbhip_selftest(2, _p(keys), n, k, _p(out)))
        perm = np.zeros(n, np.int32)
        resp = score.astype(np.float32)
        m = L.orbref_nth_prefix(_p(resp), n, k, _p(perm))
        assert m == k
        assert np.array_equal(out[:k] & 0xFFFFFF, perm[:k].astype(np.uint32)), (n, k)


@pytest.mark.parametrize("which", [4, 5])
@pytest.mark.parametrize("n,mode", [(4, 0), (5, 1), (17, 0), (64, 1), (65, 0), (100, 0), (129, 2), (480, 0), (600, 1),
                                    (1536, 0), (3000, 0), (4096, 2)])
def test_wave_cooperative_nth_element_matches_libstdcxx(orb, orbref, n, mode, which):
    """The wave-parallel partition must move elements exactly like the serial scan: whole-array
    permutation equality with std::nth_element (which 4) and with libstdc++'s __introselect at
    depth limit 1, which takes the heap-select branch (which 5; compared on the kept prefix set)."""
    rng = np.random.default_rng(1000 * which + n + mode)
    if mode == 0:
        score = rng.integers(7, 12, n)
    elif mode == 1:
        score = rng.integers(0, 255, n)
    else:
        score = np.where(rng.integers(0, 8, n) == 0, rng.integers(0, 255, n), 20)
    keys = (score.astype(np.uint32) << 24) | np.arange(n, dtype=np.uint32)
    for k in sorted({1, n // 7 + 1, n // 2, n - 1}):
        if not 0 < k < n:
            continue
        wave = np.zeros(n, np.uint32)
        orb._lib.check(orb._lib.lib().orbhip_selftest(which, _p(keys), n, k, _p(wave)))
        if which == 4:
            serial = np.zeros(n, np.uint32)
            orb._lib.check(orb._lib.lib().orbhip_selftest(2, _p(keys), n, k, _p(serial)))
            assert np.array_equal(wave, serial), (n, k)  # serial device version is itself pinned to libstdc++
            perm = np.zeros(n, np.int32)
            resp = score.astype(np.float32)
            assert orbref.lib().orbref_nth_prefix(_p(resp), n, k, _p(perm)) == k
            assert np.array_equal(wave[:k] & 0xFFFFFF, perm[:k].astype(np.uint32)), (n, k)
        else:
            assert sorted(wave & 0xFFFFFF) == list(range(n))
            kept = np.sort(score[(wave[:k] & 0xFFFFFF).astype(int)])[::-1]
            assert np.array_equal(kept, np.sort(score)[::-1][:k])


def test_corner_measure_matches_opencv_style_score(orb, orbref):
    rng = np.random.default_rng(3)
    n = 50000
    patches = rng.integers(0, 256, (n, 7, 7), dtype=np.uint8)
    patches[: n // 2] = (patches[: n // 2] // 32) * 32  # coarse levels: more corners and ties
    out = np.zeros(n, np.int32)
    orb._lib.check(orb._lib.lib().orbhip_selftest(3, _p(patches), n, 0, _p(out)))
    L = orbref.lib()
    for i in range(0, 5000):
        c = patches[i].ctypes.data + 24
        assert out[i] == L.orbref_fast_measure(c, 7)
