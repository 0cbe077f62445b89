"""Host-side mirror of the OpenCV matcher interface the reference calls.

Reference call sites (paths in the reference tree):
  src/orb_feature_matching/matcher.py:18   bf = cv.BFMatcher(cv.NORM_HAMMING, crossCheck=True)
  src/orb_feature_matching/matcher.py:21   matches = bf.match(des1, des2)
  src/orb_feature_matching/matcher.py:24   sorted(matches, key=lambda x: x.distance)
  src/cereal_box.cpp:53,344                DescriptorMatcher::create("BruteForce-Hamming")->match
  src/cereal_box.cpp:345-356               min-distance filter

Same names, argument meaning and error behaviour as cv2 for this path; the
arithmetic runs in libposehip.so on the GPU (no CPU fallback).
"""
import ctypes as C

import numpy as np

from . import _lib

NORM_HAMMING = 6  # cv2.NORM_HAMMING


class DMatch:
    """cv2.DMatch: queryIdx, trainIdx, imgIdx, distance."""
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, queryIdx=-1, trainIdx=-1, imgIdx=-1, distance=float("inf")):
        self.queryIdx = int(queryIdx)
        self.trainIdx = int(trainIdx)
        self.imgIdx = int(imgIdx)
        self.distance = float(distance)

    def __repr__(self):
        return (f"DMatch(queryIdx={self.queryIdx}, trainIdx={self.trainIdx}, "
                f"imgIdx={self.imgIdx}, distance={self.distance})")

    def __eq__(self, other):
        return (isinstance(other, DMatch) and self.queryIdx == other.queryIdx
                and self.trainIdx == other.trainIdx and self.imgIdx == other.imgIdx
                and self.distance == other.distance)


def _as_descriptors(a, name):
    """uint8 [N, 32] with unit column stride; returns (array, rows, row stride)."""
    if a is None:
        return np.zeros((0, 32), np.uint8), 0, 32
    a = np.asarray(a)
    if a.size == 0:
        return np.zeros((0, 32), np.uint8), 0, 32
    if a.dtype != np.uint8 or a.ndim != 2 or a.shape[1] != 32:
        # cv2 raises cv2.error from a CV_Assert on type/cols mismatch
        raise ValueError(f"{name}: expected uint8 array of shape [N, 32] (256-bit ORB "
                         f"descriptors), got dtype={a.dtype} shape={a.shape}")
    if a.strides[1] != 1 or a.strides[0] < 32:
        a = np.ascontiguousarray(a)
    return a, a.shape[0], a.strides[0]


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


_DM_DTYPE = np.dtype([("q", "<i4"), ("t", "<i4"), ("i", "<i4"), ("d", "<f4")])


def _dm(q, t, i, d):
    m = object.__new__(DMatch)  # the values are Python ints / floats already
    m.queryIdx, m.trainIdx, m.imgIdx, m.distance = q, t, i, d
    return m


def _dm_list(buf, n):
    if n == 0:
        return []
    a = np.frombuffer(buf, dtype=_DM_DTYPE, count=n)
    return list(map(_dm, a["q"].tolist(), a["t"].tolist(), a["i"].tolist(), a["d"].tolist()))


class BFMatcher:
    """cv2.BFMatcher restricted to NORM_HAMMING on 256-bit descriptors."""

    def __init__(self, normType=NORM_HAMMING, crossCheck=False):
        if normType != NORM_HAMMING:
            raise ValueError("only NORM_HAMMING is implemented (the reference uses no other norm)")
        self.normType = normType
        self.crossCheck = bool(crossCheck)
        lib = _lib.load()
        if lib.ph_device() < 0:
            _lib.init(0)
        self._train = []        # descriptor sets given to add(), one per train image
        self._db = None         # handle of their resident concatenation
        self._img_start = None  # first row of every train image in it

    # ---- the train collection of cv2.DescriptorMatcher --------------------------------
    def add(self, descriptors):
        """DescriptorMatcher.add: one descriptor array per train image."""
        if isinstance(descriptors, np.ndarray) and descriptors.ndim == 2:
            descriptors = [descriptors]
        for d in descriptors:
            a, n, _ = _as_descriptors(d, "descriptors")
            self._train.append(np.ascontiguousarray(a[:, :32]) if n else np.zeros((0, 32), np.uint8))
        self._release_db()

    def getTrainDescriptors(self):
        return list(self._train)

    def empty(self):
        return not any(len(t) for t in self._train)

    def clear(self):
        self._train = []
        self._release_db()

    def train(self):
        """Uploads the collection to the GPU once; match()/knnMatch() without a train
        argument then send only the query descriptors."""
        if self._db is not None:
            return
        allrows = (np.concatenate(self._train) if self._train else np.zeros((0, 32), np.uint8))
        self._img_start = np.concatenate([[0], np.cumsum([len(t) for t in self._train])]).astype(np.int64)
        h = C.c_void_p()
        _lib.check(_lib.load().ph_db_upload(_u8(allrows), allrows.shape[0], 32, C.byref(h)))
        self._db = h

    def _release_db(self):
        if getattr(self, "_db", None) is not None:
            lib = _lib.load()
            if lib.ph_db_rows(self._db) >= 0:     # not already released by ph_shutdown
                lib.ph_db_free(self._db)
        self._db = None

    def __del__(self):
        try:
            self._release_db()
        except Exception:  # interpreter shutdown
            pass

    def _collection_knn(self, queryDescriptors, k):
        """k-NN against the added collection: (idx, dist, imgIdx) arrays, idx local to the image."""
        q, nq, qs = _as_descriptors(queryDescriptors, "queryDescriptors")
        self.train()
        lib = _lib.load()
        if lib.ph_db_rows(self._db) < 0:          # the library was shut down in between
            self._db = None
            self.train()
        idx = np.full((nq, k), _lib.NO_MATCH_IDX, np.int32)
        dist = np.full((nq, k), _lib.NO_MATCH_DIST, np.int32)
        i32 = C.POINTER(C.c_int32)
        if nq and lib.ph_db_rows(self._db) > 0:
            _lib.check(lib.ph_db_knn_hamming256(self._db, _u8(q), nq, qs, k, idx.ctypes.data_as(i32),
                                                dist.ctypes.data_as(i32), None, 0.0))
        img = np.searchsorted(self._img_start, np.maximum(idx, 0), side="right") - 1
        local = idx - self._img_start[img].astype(np.int32)
        local[idx < 0] = -1
        return local, dist, img

    def _collection_match(self, queryDescriptors, k):
        if self.crossCheck:
            # the cross-check needs the swapped search as well: one call on the concatenation
            allrows = np.concatenate(self._train) if self._train else np.zeros((0, 32), np.uint8)
            self.train()
            flat = self.match(queryDescriptors, allrows)
            for m in flat:
                g = m.trainIdx
                m.imgIdx = int(np.searchsorted(self._img_start, g, side="right") - 1)
                m.trainIdx = int(g - self._img_start[m.imgIdx])
            return [[m] for m in flat]
        local, dist, img = self._collection_knn(queryDescriptors, k)
        return [[DMatch(i, int(local[i, r]), int(img[i, r]), float(dist[i, r]))
                 for r in range(k) if local[i, r] >= 0] for i in range(local.shape[0])]

    def match(self, queryDescriptors, trainDescriptors=None):
        if trainDescriptors is None:   # DescriptorMatcher.match(query): against the collection
            return [row[0] for row in self._collection_match(queryDescriptors, 1) if row]
        q, nq, qs = _as_descriptors(queryDescriptors, "queryDescriptors")
        t, nt, ts = _as_descriptors(trainDescriptors, "trainDescriptors")
        lib = _lib.load()
        out = (_lib.DMatchC * max(nq, 1))()
        n = C.c_int(0)
        fn = lib.ph_match_crosscheck_hamming256 if self.crossCheck else lib.ph_match_hamming256
        _lib.check(fn(_u8(q), nq, qs, _u8(t), nt, ts, out, C.byref(n)))
        return _dm_list(out, n.value)

    def knnMatch(self, queryDescriptors, trainDescriptors=None, k=None):
        if k is None:                  # knnMatch(query, k): against the collection
            if trainDescriptors is None or not isinstance(trainDescriptors, (int, np.integer)):
                raise TypeError("knnMatch(query, k) or knnMatch(query, train, k)")
            k, trainDescriptors = int(trainDescriptors), None
        if trainDescriptors is None:
            if self.crossCheck and k != 1:
                raise ValueError("crossCheck requires k == 1")
            if k < 1 or k > 2:
                raise ValueError("k must be 1 or 2 (the hot path uses k = 1 and k = 2)")
            return self._collection_match(queryDescriptors, k)
        if self.crossCheck and k != 1:
            # BFMatcher::knnMatchImpl asserts K == 1 when crossCheck is set
            raise ValueError("crossCheck requires k == 1")
        if k < 1:
            raise ValueError("k must be >= 1")
        if k > 2:
            raise ValueError("k > 2 is not implemented (the hot path uses k = 1 and k = 2)")
        if self.crossCheck:
            return [[m] for m in self.match(queryDescriptors, trainDescriptors)]
        q, nq, qs = _as_descriptors(queryDescriptors, "queryDescriptors")
        t, nt, ts = _as_descriptors(trainDescriptors, "trainDescriptors")
        if nq == 0 or nt == 0:
            return []
        idx = np.empty((nq, k), np.int32)
        dist = np.empty((nq, k), np.int32)
        i32 = C.POINTER(C.c_int32)
        _lib.check(_lib.load().ph_knn_hamming256(_u8(q), nq, qs, _u8(t), nt, ts, k,
                                                 idx.ctypes.data_as(i32),
                                                 dist.ctypes.data_as(i32)))
        rows = []
        for i in range(nq):
            rows.append([DMatch(i, int(idx[i, r]), 0, float(dist[i, r]))
                         for r in range(k) if idx[i, r] >= 0])
        return rows

    def knnMatchArrays(self, queryDescriptors, trainDescriptors, k):
        """knnMatch as (idx[nq,k], dist[nq,k]) int32 arrays; -1 / INT_MAX mark empty slots."""
        q, nq, qs = _as_descriptors(queryDescriptors, "queryDescriptors")
        t, nt, ts = _as_descriptors(trainDescriptors, "trainDescriptors")
        idx = np.full((nq, k), _lib.NO_MATCH_IDX, np.int32)
        dist = np.full((nq, k), _lib.NO_MATCH_DIST, np.int32)
        i32 = C.POINTER(C.c_int32)
        _lib.check(_lib.load().ph_knn_hamming256(_u8(q), nq, qs, _u8(t), nt, ts, k,
                                                 idx.ctypes.data_as(i32),
                                                 dist.ctypes.data_as(i32)))
        return idx, dist


def ratio_test_match(queryDescriptors, trainDescriptors, ratio=0.75):
    """knnMatch(k=2) + Lowe ratio test (m.distance < ratio * n.distance).

    Not in the reference; the ratio default 0.75 is this build's choice.
    """
    q, nq, qs = _as_descriptors(queryDescriptors, "queryDescriptors")
    t, nt, ts = _as_descriptors(trainDescriptors, "trainDescriptors")
    out = (_lib.DMatchC * max(nq, 1))()
    n = C.c_int(0)
    _lib.check(_lib.load().ph_knn2_ratio_hamming256(_u8(q), nq, qs, _u8(t), nt, ts,
                                                    float(ratio), out, C.byref(n)))
    return _dm_list(out, n.value)


def min_dist_filter(matches):
    """good_matches of src/cereal_box.cpp:345-356: distance <= max(2*min_dist, 30.0)."""
    n = len(matches)
    buf = (_lib.DMatchC * max(n, 1))()
    for i, m in enumerate(matches):
        buf[i] = _lib.DMatchC(m.queryIdx, m.trainIdx, m.imgIdx, m.distance)
    out = (_lib.DMatchC * max(n, 1))()
    k = C.c_int(0)
    _lib.check(_lib.load().ph_min_dist_filter(buf, n, out, C.byref(k)))
    return _dm_list(out, k.value)
