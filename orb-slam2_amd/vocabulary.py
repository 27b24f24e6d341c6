"""Host-side mirror of the part of ORB_SLAM::ORBVocabulary (include/ORBVocabulary.h, a
DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) that frames and keyframes use:
transform(features, BowVector, FeatureVector, levelsup) (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:
1125-1194) and score (:1199-1203).  The tree descent of every descriptor runs on the device
(liborbhip.so); the accumulation into the two maps follows the reference's feature order on the host.
"""
import ctypes as C

import numpy as np

from . import _lib

TF_IDF, TF, IDF, BINARY = 0, 1, 2, 3                                        # BowVector.h:36-42
L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT = range(6)    # BowVector.h:45-53


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class ORBVocabulary:
    """Tree as arrays in node-id order (node 0 = root): parent[nnodes], desc[nnodes, 32] u8,
    weight[nnodes] float64 (read at leaves), depth L."""

    def __init__(self, parent, desc, weight, L, weighting=TF_IDF, scoring=L1_NORM, device=0):
        self.parent = np.ascontiguousarray(parent, dtype=np.int32)
        self.desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, 32)
        self.weight = np.ascontiguousarray(weight, dtype=np.float64)
        assert len(self.parent) == len(self.desc) == len(self.weight)
        self.L, self.weighting, self.scoring, self.device = int(L), weighting, scoring, device
        self._L = _lib.lib()
        self._h = C.c_void_p()
        _lib.check(self._L.orbhip_vocabulary_create(C.byref(self._h), device, len(self.parent), _p(self.parent),
                                                    _p(self.desc), _p(self.weight), self.L))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.orbhip_vocabulary_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        """Number of words (TemplatedVocabulary::size)."""
        return self._L.orbhip_vocabulary_words(self._h)

    def transform_device(self, matcher, d_desc, n, levelsup, d_word, d_node, d_weight):
        """Raw device form: per descriptor the word id, the node id at level L - levelsup, the weight."""
        _lib.check(self._L.orbhip_bow_transform_device(matcher._h, self._h, d_desc, n, levelsup, d_word, d_node, d_weight))

    def transform_features(self, matcher, descriptors, levelsup=4):
        """descriptors: [n, 32] u8 host array or CUDA tensor -> (word, node, weight) host arrays."""
        import torch
        dev = "cuda:%d" % self.device
        d = descriptors if isinstance(descriptors, torch.Tensor) else torch.from_numpy(
            np.ascontiguousarray(descriptors, dtype=np.uint8).reshape(-1, 32)).to(dev)
        n = int(d.shape[0])
        word = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        node = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        wgt = torch.empty(max(n, 1), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        self.transform_device(matcher, d.data_ptr(), n, levelsup, word.data_ptr(), node.data_ptr(), wgt.data_ptr())
        matcher.synchronize()
        return word[:n].cpu().numpy(), node[:n].cpu().numpy(), wgt[:n].cpu().numpy()

    def transform(self, matcher, descriptors, levelsup=4):
        """void transform(const vector<TDescriptor>& features, BowVector &v, FeatureVector &fv, int levelsup):
        returns (v, fv) as dicts in ascending key order, values float (v) / lists of feature indices (fv)."""
        word, node, wgt = self.transform_features(matcher, descriptors, levelsup)
        v, fv = {}, {}
        accumulate = self.weighting in (TF, TF_IDF)
        for i in range(len(word)):
            w = float(wgt[i])
            if w > 0:  # "not stopped"
                k = int(word[i])
                if accumulate:
                    v[k] = v[k] + w if k in v else w  # BowVector::addWeight
                elif k not in v:
                    v[k] = w                          # BowVector::addIfNotExist
                fv.setdefault(int(node[i]), []).append(i)
        v = dict(sorted(v.items()))
        fv = dict(sorted(fv.items()))
        must = self.scoring != DOT_PRODUCT  # ScoringObject.h:72-87
        if accumulate and v and not must:
            nd = float(len(v))
            for k in v:
                v[k] /= nd
        if must:  # BowVector::normalize, sums in key order
            norm = 0.0
            if self.scoring == L2_NORM:
                for x in v.values():
                    norm += x * x
                norm = float(np.sqrt(norm))
            else:
                for x in v.values():
                    norm += abs(x)
            if norm > 0.0:
                for k in v:
                    v[k] /= norm
        return v, fv

    @staticmethod
    def score_l1(v1, v2):
        """L1Scoring::score (ScoringObject.cpp:23-68) of two BowVector dicts."""
        score = 0.0
        for k in sorted(set(v1) & set(v2)):
            vi, wi = v1[k], v2[k]
            score += abs(vi - wi) - abs(vi) - abs(wi)
        return -score / 2.0
