"""Host-side mirror of the OpenCV pose interface of the reference's EPnP branch.

The reference's report states "we implemented EPnP with Ransac" (docs/index.md:120,
also :75) and its checkout carries the TODO where the call belongs
(src/cereal_box.cpp:292-295): cv::solvePnPRansac(..., SOLVEPNP_EPNP).  Same names,
argument meaning, defaults and return tuple as cv2; the arithmetic runs in
libposehip.so on the GPU (no CPU fallback).
"""
import ctypes as C

import numpy as np

from . import _lib

SOLVEPNP_EPNP = 1  # cv2.SOLVEPNP_EPNP
CV_RNG_DEFAULT_STATE = 0xFFFFFFFFFFFFFFFF  # RNG rng((uint64)-1) inside OpenCV's RANSAC


def _points(a, cols, name):
    a = np.asarray(a)
    if a.size == 0:
        raise ValueError(f"{name}: no points")
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, cols)  # cv2 converts to CV_32F
    return a


def _camera(K, dist):
    K = np.ascontiguousarray(K, dtype=np.float64)
    if K.shape != (3, 3):
        raise ValueError("cameraMatrix must be 3x3")
    if dist is not None and np.size(dist) and np.any(np.asarray(dist, dtype=np.float64) != 0):
        raise ValueError("lens distortion is not implemented (the reference renders with an "
                         "ideal pinhole camera)")
    return K.reshape(9)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def solvePnP(objectPoints, imagePoints, cameraMatrix, distCoeffs=None, flags=SOLVEPNP_EPNP):
    """cv2.solvePnP(..., flags=SOLVEPNP_EPNP) -> (retval, rvec[3,1], tvec[3,1])."""
    if flags != SOLVEPNP_EPNP:
        raise ValueError("only SOLVEPNP_EPNP is implemented")
    obj, img = _points(objectPoints, 3, "objectPoints"), _points(imagePoints, 2, "imagePoints")
    if obj.shape[0] != img.shape[0]:
        raise ValueError("objectPoints and imagePoints differ in length")
    K = _camera(cameraMatrix, distCoeffs)
    rvec, tvec = np.zeros(3), np.zeros(3)
    _lib.check(_lib.load().ph_solvepnp_epnp(_fp(obj), _fp(img), obj.shape[0], _dp(K), _dp(rvec),
                                            _dp(tvec)))
    return True, rvec.reshape(3, 1), tvec.reshape(3, 1)


def solvePnPRansac(objectPoints, imagePoints, cameraMatrix, distCoeffs=None,
                   useExtrinsicGuess=False, iterationsCount=100, reprojectionError=8.0,
                   confidence=0.99, flags=SOLVEPNP_EPNP, rngState=CV_RNG_DEFAULT_STATE):
    """cv2.solvePnPRansac -> (retval, rvec[3,1], tvec[3,1], inliers[k,1] int32 or None).

    rngState is the one addition: OpenCV hard-codes its RANSAC seed, this keeps it
    as the default and lets a caller vary it.
    """
    if flags != SOLVEPNP_EPNP:
        raise ValueError("only SOLVEPNP_EPNP is implemented")
    if useExtrinsicGuess:
        raise ValueError("useExtrinsicGuess has no effect with SOLVEPNP_EPNP and is not accepted")
    obj, img = _points(objectPoints, 3, "objectPoints"), _points(imagePoints, 2, "imagePoints")
    if obj.shape[0] != img.shape[0]:
        raise ValueError("objectPoints and imagePoints differ in length")
    K = _camera(cameraMatrix, distCoeffs)
    n = obj.shape[0]
    rvec, tvec = np.zeros(3), np.zeros(3)
    inl = np.empty(n, np.int32)
    n_inl, found, iters = C.c_int(0), C.c_int(0), C.c_int(0)
    _lib.check(_lib.load().ph_solvepnp_ransac_epnp(
        _fp(obj), _fp(img), n, _dp(K), int(iterationsCount), float(reprojectionError),
        float(confidence), int(rngState), _dp(rvec), _dp(tvec),
        inl.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n_inl), C.byref(found), C.byref(iters)))
    if not found.value:
        return False, rvec.reshape(3, 1), tvec.reshape(3, 1), None
    return True, rvec.reshape(3, 1), tvec.reshape(3, 1), inl[:n_inl.value].reshape(-1, 1).copy()


def solvePnPRansacBatch(objectPointsList, imagePointsList, cameraMatrix, distCoeffs=None,
                        iterationsCount=100, reprojectionError=8.0, confidence=0.99,
                        flags=SOLVEPNP_EPNP, rngState=CV_RNG_DEFAULT_STATE,
                        returnIterations=False):
    """solvePnPRansac for a batch of frames in one call: element p of the returned
    list is what solvePnPRansac(objectPointsList[p], imagePointsList[p], ...) returns
    (plus the RANSAC iteration count with returnIterations).  A frame with fewer than
    5 correspondences yields (False, zeros, zeros, None).
    """
    if flags != SOLVEPNP_EPNP:
        raise ValueError("only SOLVEPNP_EPNP is implemented")
    if len(objectPointsList) != len(imagePointsList):
        raise ValueError("objectPointsList and imagePointsList differ in length")
    K = _camera(cameraMatrix, distCoeffs)
    P = len(objectPointsList)
    if P == 0:
        return []
    # pack every frame's points into one buffer each (np.concatenate converts to float32 while
    # it copies; frames given as (n, 1, 3) / (n, 1, 2), as cv2 accepts them, are flattened first)
    objs = [o if getattr(o, "ndim", 0) == 2 else np.asarray(o).reshape(-1, 3) for o in objectPointsList]
    imgs = [i if getattr(i, "ndim", 0) == 2 else np.asarray(i).reshape(-1, 2) for i in imagePointsList]
    counts = [o.shape[0] for o in objs]
    if counts != [i.shape[0] for i in imgs]:
        raise ValueError("objectPoints and imagePoints of a frame differ in length")
    if any(o.shape[1] != 3 for o in objs) or any(i.shape[1] != 2 for i in imgs):
        raise ValueError("objectPoints must be (n, 3) and imagePoints (n, 2)")
    off = np.zeros(P + 1, np.int32)
    np.cumsum(counts, out=off[1:])
    total = int(off[-1])
    obj = np.concatenate(objs, dtype=np.float32, casting="unsafe") if total else np.zeros((1, 3), np.float32)
    img = np.concatenate(imgs, dtype=np.float32, casting="unsafe") if total else np.zeros((1, 2), np.float32)
    rvecs, tvecs = np.zeros((P, 3, 1)), np.zeros((P, 3, 1))
    inl = np.zeros((max(total, 1), 1), np.int32)
    n_inl, found, iters = (np.zeros(P, np.int32) for _ in range(3))
    i32 = C.POINTER(C.c_int32)
    _lib.check(_lib.load().ph_solvepnp_ransac_epnp_batched(
        _fp(obj), _fp(img), off.ctypes.data_as(i32), P, _dp(K), int(iterationsCount),
        float(reprojectionError), float(confidence), int(rngState), _dp(rvecs), _dp(tvecs),
        inl.ctypes.data_as(i32), n_inl.ctypes.data_as(i32), found.ctypes.data_as(i32),
        iters.ctypes.data_as(i32)))
    # the results are views of this call's own arrays (nothing of the library's is aliased)
    lo, hi = off[:-1].tolist(), (off[:-1] + n_inl).tolist()
    ok, its = found.astype(bool).tolist(), iters.tolist()
    if returnIterations:
        return [(ok[p], rvecs[p], tvecs[p], inl[lo[p]:hi[p]] if ok[p] else None, its[p]) for p in range(P)]
    return [(ok[p], rvecs[p], tvecs[p], inl[lo[p]:hi[p]] if ok[p] else None) for p in range(P)]
