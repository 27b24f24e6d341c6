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


def _frames(orbref, rng, desc_pool, sizes):
    out = []
    for n in sizes:
        kp = np.zeros(n, orbref.KP_DTYPE)
        kp["x"], kp["y"] = rng.uniform(0, 640, n), rng.uniform(0, 480, n)
        kp["angle"] = rng.uniform(0, 360, n).astype(np.float32)
        out.append((kp, _features(rng, desc_pool, n) if n else np.zeros((0, 32), np.uint8)))
    return out


def _second_view(rng, kp, de, rot=25.0):
    keep = rng.random(len(kp)) > 0.1
    kp2, de2 = kp[keep].copy(), de[keep].copy()
    kp2["angle"] = np.mod(kp2["angle"] - np.float32(rot) + rng.normal(0, 6, len(kp2)).astype(np.float32), 360).astype(np.float32)
    wild = rng.random(len(kp2)) < 0.1
    kp2["angle"][wild] = rng.uniform(0, 360, int(wild.sum())).astype(np.float32)
    for i in range(len(kp2)):
        for _ in range(rng.integers(0, 6)):
            de2[i, rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
    p = rng.permutation(len(kp2))
    return kp2[p], de2[p]


@pytest.mark.parametrize("kf_frame", [False, True])
@pytest.mark.parametrize("ori", [True, False])
def test_search_by_bow(orb, orbref, synth, kf_frame, ori):
    import torch
    k, L, levelsup = 10, 4, 2   # FeatureVector nodes at level 2: up to 100 nodes
    parent, desc, weight = synth.synth_vocabulary(k, L, seed=17)
    voc = orb.ORBVocabulary(parent, desc, weight, L)
    ref = orbref.RefVocabulary(parent, desc, weight, L)
    rng = np.random.default_rng(23)
    fr1 = _frames(orbref, rng, desc, [0, 3, 200, 1500, 2000])
    fr2 = [_second_view(rng, kp, de) for kp, de in fr1]
    fr2[2] = (fr2[2][0][:0], fr2[2][1][:0])  # an empty second frame
    m = orb.ORBmatcher(0.8 if not kf_frame else 0.7, ori)
    cap = max(max(len(a[0]), len(b[0])) for a, b in zip(fr1, fr2)) or 1
    F1 = orb.FrameBatch.from_host(m, [a for a, _ in fr1], [b for _, b in fr1], [0, 0, 640, 480], cap=cap)
    F2 = orb.FrameBatch.from_host(m, [a for a, _ in fr2], [b for _, b in fr2], [0, 0, 640, 480], cap=cap)
    F1.ComputeBoW(voc, levelsup, node_cap=256)
    F2.ComputeBoW(voc, levelsup, node_cap=256)
    v1 = np.zeros((len(fr1), cap), np.uint8)
    v2 = np.zeros((len(fr1), cap), np.uint8)
    for f in range(len(fr1)):
        v1[f, :len(fr1[f][0])] = rng.random(len(fr1[f][0])) < 0.8
        v2[f, :len(fr2[f][0])] = rng.random(len(fr2[f][0])) < 0.9
    nm, m12, m21 = m.SearchByBoW(F1, F2, torch.from_numpy(v1).cuda(), torch.from_numpy(v2).cuda(), kf_frame=kf_frame)
    nm, m12, m21 = nm.cpu().numpy(), m12.cpu().numpy(), m21.cpu().numpy()
    total = 0
    for f in range(len(fr1)):
        (kp1, de1), (kp2, de2) = fr1[f], fr2[f]
        _, fv1 = ref.transform(de1, levelsup)
        _, fv2 = ref.transform(de2, levelsup)
        # the device FeatureVector equals the reference's
        for F, fv, nfeat in ((F1, fv1, len(kp1)), (F2, fv2, len(kp2))):
            dn, ds, dfeat = F.feature_vector(f)
            assert np.array_equal(dn, fv[0]) and np.array_equal(ds, fv[1]) and np.array_equal(dfeat, fv[2]), "FeatureVector %d" % f
        if kf_frame:
            rn, rF = orbref.search_by_bow_kf_frame(kp1, de1, v1[f, :len(kp1)], fv1, kp2, de2, fv2, m.mfNNratio, ori)
            assert np.array_equal(rF, m21[f, :len(kp2)]), "pair %d vpMapPointMatches" % f
        else:
            rn, r12 = orbref.search_by_bow_kf_kf(kp1, de1, v1[f, :len(kp1)], fv1, kp2, de2, v2[f, :len(kp2)], fv2,
                                                 m.mfNNratio, ori)
            assert np.array_equal(r12, m12[f, :len(kp1)]), "pair %d vpMatches12" % f
        assert rn == nm[f], "pair %d nmatches" % f
        # the two output arrays describe the same matching
        for i1 in np.nonzero(m12[f] >= 0)[0]:
            assert m21[f, m12[f, i1]] == i1
        assert (m21[f] >= 0).sum() == (m12[f] >= 0).sum() == nm[f]
        total += rn
    assert total > 300


def test_vocabulary_text_file_round_trip(orb, orbref, synth, tmp_path):
    """ORBvoc.txt format (TemplatedVocabulary.h:1337-1445): save, reload, same tree and same transform."""
    k, L = 5, 3
    parent, desc, weight = synth.synth_vocabulary(k, L, seed=29)
    voc = orb.ORBVocabulary(parent, desc, weight, L, weighting=1, scoring=0)
    path = str(tmp_path / "voc.txt")
    voc.saveToTextFile(path, k=k)
    head = open(path).readline().split()
    assert head == [str(k), str(L), "0", "1"]
    v2 = orb.ORBVocabulary.loadFromTextFile(path)
    # the root (node 0) is not in the file: its descriptor and weight are never read
    assert np.array_equal(v2.parent, voc.parent) and np.array_equal(v2.desc[1:], voc.desc[1:])
    assert np.array_equal(v2.weight[1:], voc.weight[1:])
    assert (v2.L, v2.weighting, v2.scoring, v2.size()) == (L, 1, 0, voc.size())
    m = orb.ORBmatcher()
    rng = np.random.default_rng(31)
    f = _features(rng, desc, 500)
    a, b = voc.transform_features(m, f, 2), v2.transform_features(m, f, 2)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    ref = orbref.RefVocabulary(v2.parent, v2.desc, v2.weight, L, weighting=1, scoring=0)
    assert all(np.array_equal(x, y) for x, y in zip(b, ref.transform_features(f, 2)))
    # a file that fails the header check (src/.../TemplatedVocabulary.h:1358-1362) is refused
    bad = tmp_path / "bad.txt"
    bad.write_text("25 3 0 0\n")
    assert orb.ORBVocabulary.loadFromTextFile(str(bad)) is None


def test_vocabulary_of_the_real_shape_k10_L6(orb, orbref, synth, tmp_path):
    """The shape of the vocabulary the reference loads (ORBvoc: k = 10, L = 6, 1,111,111 nodes, 1 M words,
    Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1337-1420) on a synthetic tree of that size: the transform with the
    levelsup Frame::ComputeBoW uses (4) against the oracle, then the tree through the ORBvoc text format (155 MB) and
    back, same arrays and same transform."""
    k, L = 10, 6
    parent, desc, weight = synth.synth_vocabulary_large(k, L, seed=61)
    assert len(parent) == 1111111
    voc = orb.ORBVocabulary(parent, desc, weight, L)
    assert voc.size() == 10 ** 6
    ref = orbref.RefVocabulary(parent, desc, weight, L)
    m = orb.ORBmatcher()
    rng = np.random.default_rng(67)
    n = 20000
    f = desc[rng.integers(1, len(desc), n)] ^ (rng.integers(0, 256, (n, 32), dtype=np.uint8)
                                                & rng.integers(0, 256, (n, 32), dtype=np.uint8)
                                                & rng.integers(0, 256, (n, 32), dtype=np.uint8))
    f[: n // 10] = rng.integers(0, 256, (n // 10, 32), dtype=np.uint8)
    got = voc.transform_features(m, f, 4)
    want = ref.transform_features(f, 4)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal(got[2].view(np.uint64), want[2].view(np.uint64))
    # node ids at level L - 4 = 2: the hundred grandchildren of the root
    assert len(np.unique(got[1])) <= 100 and len(np.unique(got[0])) > n // 2
    path = str(tmp_path / "voc_k10_L6.txt")
    voc.saveToTextFile(path, k=k)
    v2 = orb.ORBVocabulary.loadFromTextFile(path)
    assert (v2.k, v2.L, v2.size()) == (k, L, 10 ** 6)
    assert np.array_equal(v2.parent, parent) and np.array_equal(v2.desc[1:], desc[1:]) and np.array_equal(v2.weight[1:], weight[1:])
    again = v2.transform_features(m, f[:4000], 4)
    assert all(np.array_equal(a, b[:4000]) for a, b in zip(again, got))
