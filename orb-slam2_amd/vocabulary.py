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

    @classmethod
    def loadFromTextFile(cls, filename, device=0):
        """The text format written by TemplatedVocabulary::saveToTextFile (TemplatedVocabulary.h:1424-1445) and read
        by loadFromTextFile (:1337-1420), the format of ORBvoc.txt: a header "k L scoring weighting", then one line per
        node in id order (the root has none): "parent isLeaf d0 ... d31 weight".  Nodes are numbered by line, children
        attach in line order and words number the leaves in line order, as the reference does.  All 32 descriptor bytes
        are read (the reference's v1 loader keeps only the last token of a line's descriptor, :1392-1394; a file it
        wrote itself reloads here as saved).  Returns the vocabulary, or None for a file that fails the header check."""
        with open(filename, "r") as f:
            head = f.readline().split()
            if len(head) < 4:
                return None
            k, L, scoring, weighting = (int(v) for v in head[:4])
            if k < 0 or k > 20 or L < 1 or L > 10 or scoring < 0 or scoring > 5 or weighting < 0 or weighting > 3:
                return None
            body = f.read()
        nrows = sum(1 for ln in body.splitlines() if ln.strip()) if len(body) < (1 << 16) else body.count("\n") + (not body.endswith("\n"))
        # every number of the node lines in one pass (the real vocabulary has 1.1 M lines of 35 fields); integers up to
        # 2^53 and the weights' decimal strings convert to the same doubles as a per-field conversion would give
        vals = np.fromstring(body, dtype=np.float64, sep=" ") if body.strip() else np.zeros(0)
        if nrows and vals.size == nrows * 35:
            tab = vals.reshape(nrows, 35)
        else:  # not the regular shape: find the line that is wrong
            rows = [ln.split() for ln in body.splitlines() if ln.strip()]
            for i, t in enumerate(rows, start=1):
                if len(t) != 35:
                    raise ValueError("%s: node line %d has %d fields, expected 35" % (filename, i, len(t)))
            tab = np.array([[float(v) for v in t] for t in rows], np.float64).reshape(len(rows), 35)
        n = len(tab) + 1
        parent = np.zeros(n, np.int32)
        desc = np.zeros((n, 32), np.uint8)
        weight = np.zeros(n, np.float64)
        leaf = np.zeros(n, bool)
        parent[1:] = tab[:, 0].astype(np.int32)
        wrong = np.nonzero((parent[1:] < 0) | (parent[1:] >= np.arange(1, n)))[0]
        if len(wrong):
            raise ValueError("%s: node %d names parent %d" % (filename, wrong[0] + 1, parent[wrong[0] + 1]))
        leaf[1:] = tab[:, 1] > 0
        desc[1:] = tab[:, 2:34].astype(np.uint8)
        weight[1:] = tab[:, 34]
        has_child = np.zeros(n, bool)
        has_child[parent[1:]] = True
        if (leaf[1:] == has_child[1:]).any():
            raise ValueError("%s: the isLeaf flags do not match the tree" % filename)
        voc = cls(parent, desc, weight, L, weighting=weighting, scoring=scoring, device=device)
        voc.k = k
        return voc

    def saveToTextFile(self, filename, k=None):
        """TemplatedVocabulary::saveToTextFile (:1424-1445)."""
        n = len(self.parent)
        has_child = np.zeros(n, bool)
        has_child[self.parent[1:]] = True
        kk = k if k is not None else getattr(self, "k", int(np.bincount(self.parent[1:]).max()) if n > 1 else 0)
        lut = np.array([str(i) for i in range(256)], dtype=object)
        with open(filename, "w") as f:
            f.write("%d %d  %d %d\n" % (kk, self.L, self.scoring, self.weighting))
            for a in range(1, n, 65536):  # in blocks: the byte -> text conversion is a table lookup per block
                b = min(n, a + 65536)
                txt = lut[self.desc[a:b]]
                par, lf, wt = self.parent[a:b].tolist(), (~has_child[a:b]).astype(int).tolist(), self.weight[a:b].tolist()
                f.write("".join("%d %d %s %r\n" % (par[i], lf[i], " ".join(txt[i]), wt[i]) for i in range(b - a)))

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
