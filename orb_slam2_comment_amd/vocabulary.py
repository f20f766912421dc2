"""Host-side mirror of ORB_SLAM2::ORBVocabulary (include/ORBVocabulary.h:32-33 =
DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) for the part the front-end calls: loadFromTextFile
(Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1424) and transform (:1127-1199), i.e. what
Frame::ComputeBoW (src/Frame.cc:395-402) needs.  Thin wrapper over the C ABI; the work happens in
csrc/orbhip_vocabulary.hip."""
import ctypes as C

import numpy as np

from .capi import NO_NODE, check, lib, ptr

# DBoW2 enums (Thirdparty/DBoW2/DBoW2/BowVector.h:36-53)
TF_IDF, TF, IDF, BINARY = 0, 1, 2, 3
L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT = 0, 1, 2, 3, 4, 5


class ORBVocabulary:
    def __init__(self, device=0):
        self._lib = lib()
        self._h = C.c_void_p()
        self._device = device

    def _replace(self, h):
        self.close()
        self._h = h

    def loadFromTextFile(self, filename):
        """Returns True on success, False (with the handle left empty) on a malformed or missing file, like the
        reference (src/System.cc:69-76 checks the flag)."""
        h = C.c_void_p()
        rc = self._lib.orbhip_vocabulary_load_text(str(filename).encode(), self._device, C.byref(h))
        if rc != 0:
            return False
        self._replace(h)
        return True

    @classmethod
    def from_arrays(cls, k, L, scoring, weighting, parent, is_leaf, desc, weight, device=0):
        """Tree from arrays: entry i describes node i+1 (file order), parent[i] in [0, i]."""
        self = cls(device)
        par = np.ascontiguousarray(parent, np.int32)
        leaf = np.ascontiguousarray(is_leaf, np.uint8)
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        w = np.ascontiguousarray(weight, np.float64)
        if not (len(par) == len(leaf) == len(d) == len(w)):
            raise ValueError("parent / is_leaf / desc / weight must have one entry per node")
        h = C.c_void_p()
        check(self._lib.orbhip_vocabulary_create(k, L, scoring, weighting, len(par), ptr(par), ptr(leaf), ptr(d), ptr(w),
                                                 device, C.byref(h)), "orbhip_vocabulary_create")
        self._h = h
        return self

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.orbhip_vocabulary_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _info(self):
        v = [C.c_int(0) for _ in range(6)]
        check(self._lib.orbhip_vocabulary_info(self._h, *[C.byref(x) for x in v]), "orbhip_vocabulary_info")
        return [x.value for x in v]

    def empty(self):
        return (not self._h) or self._info()[5] == 0

    def size(self):                       # number of words
        return 0 if not self._h else self._info()[5]

    def getBranchingFactor(self):
        return self._info()[0]

    def getDepthLevels(self):
        return self._info()[1]

    def getScoringType(self):
        return self._info()[2]

    def getWeightingType(self):
        return self._info()[3]

    def transform(self, descriptors, levelsup=4):
        """transform(features, BowVector&, FeatureVector&, levelsup).  Returns a dict with
        word_id[n], word_weight[n], node_id[n] (capi.NO_NODE for stopped words), bow_ids[m] ascending, bow_vals[m]."""
        if not self._h:
            raise RuntimeError("vocabulary not loaded")
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        n = len(d)
        word = np.zeros(max(n, 1), np.uint32)
        wgt = np.zeros(max(n, 1), np.float64)
        node = np.full(max(n, 1), NO_NODE, np.uint32)
        bid = np.zeros(max(n, 1), np.uint32)
        bval = np.zeros(max(n, 1), np.float64)
        nb = C.c_int(0)
        check(self._lib.orbhip_vocabulary_transform(self._h, ptr(d), n, int(levelsup), ptr(word), ptr(wgt), ptr(node),
                                                    ptr(bid), ptr(bval), C.byref(nb)), "orbhip_vocabulary_transform")
        return {"word_id": word[:n].copy(), "word_weight": wgt[:n].copy(), "node_id": node[:n].copy(),
                "bow_ids": bid[:nb.value].copy(), "bow_vals": bval[:nb.value].copy()}

    @staticmethod
    def feature_vector(node_id):
        """DBoW2::FeatureVector as {node: [feature indices ascending]} from the per-feature node ids."""
        fv = {}
        for i, nd in enumerate(np.asarray(node_id)):
            if nd != NO_NODE:
                fv.setdefault(int(nd), []).append(i)
        return dict(sorted(fv.items()))

    # -- device-resident, batched ------------------------------------------------
    def set_stream(self, stream):
        check(self._lib.orbhip_vocabulary_set_stream(self._h, stream), "orbhip_vocabulary_set_stream")

    def sync(self):
        check(self._lib.orbhip_vocabulary_sync(self._h), "orbhip_vocabulary_sync")

    def transform_device(self, frames, d_desc, d_n, cap, levelsup, d_word, d_weight, d_node, d_bow_ids, d_bow_vals,
                         d_n_bow):
        """All d_* are device pointers (ints) in the extractor's batch layout; asynchronous."""
        check(self._lib.orbhip_vocabulary_transform_device(self._h, frames, d_desc, d_n, cap, int(levelsup), d_word,
                                                           d_weight, d_node, d_bow_ids, d_bow_vals, d_n_bow),
              "orbhip_vocabulary_transform_device")
