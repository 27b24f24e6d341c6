"""Parity of the device bag-of-words transform against the oracle's literal restatement of
TemplatedVocabulary::transform (oracle/bow_ref.cpp), on synthetic vocabulary trees (the reference's
vocabulary file is not part of its checkout).  Bar: word ids, node ids and weights bit-exact; the
accumulated BowVector values equal as IEEE doubles."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _features(rng, desc, n):
    """Descriptors near random tree nodes, plus pure noise."""
    f = desc[rng.integers(1, len(desc), n)].copy()
    for i in range(n):
        for _ in range(rng.integers(0, 30)):
            f[i, rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
    f[: n // 10] = rng.integers(0, 256, (n // 10, 32), dtype=np.uint8)
    return f


@pytest.mark.parametrize("k,L,levelsup", [(10, 3, 1), (10, 4, 4), (10, 4, 2), (3, 6, 4), (17, 2, 1), (2, 9, 4), (10, 3, 0)])
def test_transform_features(orb, orbref, synth, k, L, levelsup):
    parent, desc, weight = synth.synth_vocabulary(k, L, seed=k * 31 + L)
    voc = orb.ORBVocabulary(parent, desc, weight, L)
    ref = orbref.RefVocabulary(parent, desc, weight, L)
    m = orb.ORBmatcher()
    rng = np.random.default_rng(7)
    f = _features(rng, desc, 1500)
    word, node, wgt = voc.transform_features(m, f, levelsup)
    rword, rnode, rwgt = ref.transform_features(f, levelsup)
    assert np.array_equal(word, rword)
    assert np.array_equal(node, rnode)
    assert np.array_equal(wgt.view(np.uint64), rwgt.view(np.uint64))
    assert voc.size() == int((np.bincount(parent[1:], minlength=len(parent)) == 0).sum())


@pytest.mark.parametrize("weighting,scoring", [(0, 0), (1, 0), (2, 1), (3, 5), (0, 5), (0, 1)])
def test_bow_and_feature_vectors(orb, orbref, synth, weighting, scoring):
    k, L = 10, 4
    parent, desc, weight = synth.synth_vocabulary(k, L, seed=99)
    voc = orb.ORBVocabulary(parent, desc, weight, L, weighting, scoring)
    ref = orbref.RefVocabulary(parent, desc, weight, L, weighting, scoring)
    m = orb.ORBmatcher()
    rng = np.random.default_rng(11)
    for n in (0, 1, 37, 2000):
        f = _features(rng, desc, n) if n else np.zeros((0, 32), np.uint8)
        v, fv = voc.transform(m, f, 2)
        (rw, rv), (fn, fs, ff) = ref.transform(f, 2)
        assert list(v.keys()) == list(rw)
        assert np.array_equal(np.array(list(v.values()), np.float64).view(np.uint64), rv.view(np.uint64))
        assert list(fv.keys()) == list(fn)
        for j, node in enumerate(fn):
            assert fv[int(node)] == list(ff[fs[j]:fs[j + 1]])


def test_l1_score_of_two_views(orb, orbref, synth):
    parent, desc, weight = synth.synth_vocabulary(10, 3, seed=5)
    voc = orb.ORBVocabulary(parent, desc, weight, 3)
    ref = orbref.RefVocabulary(parent, desc, weight, 3)
    m = orb.ORBmatcher()
    rng = np.random.default_rng(3)
    a = _features(rng, desc, 800)
    b = a.copy()
    b[::3] = _features(rng, desc, len(b[::3]))
    va, _ = voc.transform(m, a, 1)
    vb, _ = voc.transform(m, b, 1)
    ra, _ = ref.transform(a, 1)
    rb, _ = ref.transform(b, 1)
    s = voc.score_l1(va, vb)
    assert s == orbref.bow_score_l1(ra, rb)
    assert 0.0 < s < 1.0 and voc.score_l1(va, va) == pytest.approx(1.0)
