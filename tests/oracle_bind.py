"""ctypes binding of oracle/liboracle.so -- the CPU checker used by the tests.

Test infrastructure only: nothing under poseestimation_amd/ imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")


class DMatchC(C.Structure):
    _fields_ = [("queryIdx", C.c_int32), ("trainIdx", C.c_int32), ("imgIdx", C.c_int32),
                ("distance", C.c_float)]


_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_dmp = C.POINTER(DMatchC)
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith(".c")]
    stale = (not os.path.exists(LIB_PATH)
             or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs))
    if stale:
        try:
            subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
        except (OSError, subprocess.CalledProcessError):
            if not os.path.exists(LIB_PATH):
                raise
    lib = C.CDLL(LIB_PATH)
    lib.orc_knn_hamming256.restype = C.c_int
    lib.orc_knn_hamming256.argtypes = [_u8p, C.c_int, C.c_size_t, _u8p, C.c_int, C.c_size_t,
                                       C.c_int, _i32p, _i32p]
    lib.orc_knn_hamming256_mt.restype = C.c_int
    lib.orc_knn_hamming256_mt.argtypes = lib.orc_knn_hamming256.argtypes + [C.c_int]
    for name in ("orc_match", "orc_match_crosscheck"):
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = [_u8p, C.c_int, C.c_size_t, _u8p, C.c_int, C.c_size_t, _dmp]
    lib.orc_ratio_filter.restype = C.c_int
    lib.orc_ratio_filter.argtypes = [_i32p, _i32p, C.c_int, C.c_float, _dmp]
    lib.orc_min_dist_filter.restype = C.c_int
    lib.orc_min_dist_filter.argtypes = [_dmp, C.c_int, _dmp]
    lib.orc_sort_by_distance.restype = None
    lib.orc_sort_by_distance.argtypes = [_dmp, C.c_int]
    _f32p, _f64p = C.POINTER(C.c_float), C.POINTER(C.c_double)
    lib.orc_ransac_subsets.restype = None
    lib.orc_ransac_subsets.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, _i32p]
    lib.orc_ransac_update_num_iters.restype = C.c_int
    lib.orc_ransac_update_num_iters.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]
    lib.orc_rodrigues_mat_to_vec.restype = None
    lib.orc_rodrigues_mat_to_vec.argtypes = [_f64p, _f64p]
    lib.orc_rodrigues_vec_to_mat.restype = None
    lib.orc_rodrigues_vec_to_mat.argtypes = [_f64p, _f64p]
    lib.orc_solvepnp_epnp.restype = C.c_int
    lib.orc_solvepnp_epnp.argtypes = [_f32p, _f32p, C.c_int, _f64p, _f64p, _f64p]
    lib.orc_solvepnp_epnp_f64points.restype = C.c_int
    lib.orc_solvepnp_epnp_f64points.argtypes = [_f32p, _f32p, C.c_int, _f64p, _f64p, _f64p]
    lib.orc_pnp_count_inliers.restype = C.c_int
    lib.orc_pnp_count_inliers.argtypes = [_f32p, _f32p, C.c_int, _f64p, _f64p, _f64p, C.c_double,
                                          _u8p, _f32p]
    lib.orc_pnp_hypotheses.restype = C.c_int
    lib.orc_pnp_hypotheses.argtypes = [_f32p, _f32p, C.c_int, _f64p, _i32p, C.c_int, C.c_double,
                                       _f64p, _i32p, C.c_int]
    lib.orc_solvepnp_ransac.restype = C.c_int
    lib.orc_solvepnp_ransac.argtypes = [_f32p, _f32p, C.c_int, _f64p, C.c_int, C.c_float,
                                        C.c_double, C.c_uint64, _f64p, _f64p, _i32p,
                                        C.POINTER(C.c_int), C.POINTER(C.c_int)]
    _lib = lib
    return lib


def _rows(a):
    a = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1, 32) if np.size(a) else \
        np.zeros((0, 32), np.uint8)
    return a, a.shape[0]


def _p(a):
    return a.ctypes.data_as(_u8p)


def knn(q, t, k, threads=1):
    """(idx[nq,k], dist[nq,k]) int32; -1 / INT_MAX mark empty slots."""
    q, nq = _rows(q)
    t, nt = _rows(t)
    idx = np.full((nq, k), -1, np.int32)
    dist = np.full((nq, k), 2147483647, np.int32)
    lib = load()
    if threads > 1:
        rc = lib.orc_knn_hamming256_mt(_p(q), nq, 32, _p(t), nt, 32, k,
                                       idx.ctypes.data_as(_i32p), dist.ctypes.data_as(_i32p),
                                       threads)
    else:
        rc = lib.orc_knn_hamming256(_p(q), nq, 32, _p(t), nt, 32, k, idx.ctypes.data_as(_i32p),
                                    dist.ctypes.data_as(_i32p))
    assert rc == 0
    return idx, dist


def _tuples(buf, n):
    return [(m.queryIdx, m.trainIdx, m.imgIdx, m.distance) for m in buf[:n]]


def match(q, t, cross_check=False):
    """List of (queryIdx, trainIdx, imgIdx, distance)."""
    q, nq = _rows(q)
    t, nt = _rows(t)
    out = (DMatchC * max(nq, 1))()
    fn = load().orc_match_crosscheck if cross_check else load().orc_match
    n = fn(_p(q), nq, 32, _p(t), nt, 32, out)
    assert n >= 0
    return _tuples(out, n)


def ratio_filter(idx, dist, ratio):
    idx = np.ascontiguousarray(idx, np.int32)
    dist = np.ascontiguousarray(dist, np.int32)
    nq = idx.shape[0]
    out = (DMatchC * max(nq, 1))()
    n = load().orc_ratio_filter(idx.ctypes.data_as(_i32p), dist.ctypes.data_as(_i32p), nq,
                                float(ratio), out)
    return _tuples(out, n)


def min_dist_filter(matches):
    n = len(matches)
    buf = (DMatchC * max(n, 1))(*[DMatchC(*m) for m in matches])
    out = (DMatchC * max(n, 1))()
    k = load().orc_min_dist_filter(buf, n, out)
    return _tuples(out, k)


def sort_by_distance(matches):
    n = len(matches)
    buf = (DMatchC * max(n, 1))(*[DMatchC(*m) for m in matches])
    load().orc_sort_by_distance(buf, n)
    return _tuples(buf, n)


# ---- RANSAC-EPnP -------------------------------------------------------------
CV_RNG_DEFAULT = 0xFFFFFFFFFFFFFFFF  # RNG rng((uint64)-1) in RANSACPointSetRegistrator::run


def _f32(a, cols):
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1, cols)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _K(K):
    return np.ascontiguousarray(K, dtype=np.float64).reshape(9)


def ransac_subsets(n_points, n_subsets, rng_state=CV_RNG_DEFAULT, model_points=5):
    out = np.empty((n_subsets, model_points), np.int32)
    load().orc_ransac_subsets(rng_state, n_points, model_points, n_subsets,
                              out.ctypes.data_as(_i32p))
    return out


def update_num_iters(p, ep, model_points, max_iters):
    return load().orc_ransac_update_num_iters(p, ep, model_points, max_iters)


def rodrigues(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    if x.size == 9:
        r = np.empty(3)
        load().orc_rodrigues_mat_to_vec(_dp(x.reshape(9)), _dp(r))
        return r
    R = np.empty(9)
    load().orc_rodrigues_vec_to_mat(_dp(x.reshape(3)), _dp(R))
    return R.reshape(3, 3)


def solvepnp_epnp(obj, img, K, points_as_f64=False):
    """solvePnP(SOLVEPNP_EPNP) on float points; points_as_f64: the same values handed over as
    CV_64F, as solvePnPRansac does with its consensus set."""
    obj, img, K = _f32(obj, 3), _f32(img, 2), _K(K)
    r, t = np.empty(3), np.empty(3)
    fn = load().orc_solvepnp_epnp_f64points if points_as_f64 else load().orc_solvepnp_epnp
    rc = fn(_fp(obj), _fp(img), obj.shape[0], _dp(K), _dp(r), _dp(t))
    assert rc == 0
    return r, t


def count_inliers(obj, img, K, rvec, tvec, threshold=8.0):
    obj, img, K = _f32(obj, 3), _f32(img, 2), _K(K)
    n = obj.shape[0]
    mask = np.empty(n, np.uint8)
    err = np.empty(n, np.float32)
    rvec = np.ascontiguousarray(rvec, np.float64)
    tvec = np.ascontiguousarray(tvec, np.float64)
    good = load().orc_pnp_count_inliers(_fp(obj), _fp(img), n, _dp(K), _dp(rvec), _dp(tvec),
                                        threshold, mask.ctypes.data_as(_u8p), _fp(err))
    return good, mask, err


def pnp_hypotheses(obj, img, K, subsets, threshold=8.0, threads=1):
    """(models[n_hyp, 6] = rvec|tvec, counts[n_hyp])."""
    obj, img, K = _f32(obj, 3), _f32(img, 2), _K(K)
    subsets = np.ascontiguousarray(subsets, np.int32)
    h = subsets.shape[0]
    models = np.empty((h, 6), np.float64)
    counts = np.empty(h, np.int32)
    rc = load().orc_pnp_hypotheses(_fp(obj), _fp(img), obj.shape[0], _dp(K),
                                   subsets.ctypes.data_as(_i32p), h, threshold, _dp(models),
                                   counts.ctypes.data_as(_i32p), threads)
    assert rc == 0
    return models, counts


def solvepnp_ransac(obj, img, K, iterations=100, reproj_err=8.0, confidence=0.99,
                    rng_state=CV_RNG_DEFAULT):
    """(ok, rvec, tvec, inliers, iterations_run)."""
    obj, img, K = _f32(obj, 3), _f32(img, 2), _K(K)
    n = obj.shape[0]
    r, t = np.zeros(3), np.zeros(3)
    inl = np.empty(max(n, 1), np.int32)
    n_inl, iters = C.c_int(0), C.c_int(0)
    rc = load().orc_solvepnp_ransac(_fp(obj), _fp(img), n, _dp(K), iterations, reproj_err,
                                    confidence, rng_state, _dp(r), _dp(t),
                                    inl.ctypes.data_as(_i32p), C.byref(n_inl), C.byref(iters))
    return rc == 1, r, t, inl[:n_inl.value].copy(), iters.value
