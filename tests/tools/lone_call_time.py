"""Where a lone solvePnPRansac call (BASELINE configs[0] size: 500 pairs, 100 iterations) spends
its time: through the Python mirror, through the C ABI with prepared arrays, and the GPU-side
kernels of the same calls (profiled with HIP events through ph_set_profiling where available).
Prints one JSON line."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import datagen
import poseestimation_amd as pa
from poseestimation_amd import _lib

_lib.init(0)
lib = _lib.load()
n, iters = 500, 100
frames = [datagen.pnp_scene(n=n, outlier_frac=0.3, seed=500 + f) for f in range(64)]
K = np.ascontiguousarray(datagen.CAMERA_K, np.float64).reshape(9)
fp, dp, ip = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int32)


def through_python(reps):
    t0 = time.perf_counter()
    for r in range(reps):
        f = frames[r % len(frames)]
        pa.solvePnPRansac(f[0], f[1], datagen.CAMERA_K, None, iterationsCount=iters)
    return (time.perf_counter() - t0) / reps


def through_c_abi(reps):
    rvec, tvec = np.zeros(3), np.zeros(3)
    inl = np.empty(n, np.int32)
    n_inl, found, it = C.c_int(0), C.c_int(0), C.c_int(0)
    args = [(f[0].ctypes.data_as(fp), f[1].ctypes.data_as(fp)) for f in frames]
    Kp, rp, tp, lp = K.ctypes.data_as(dp), rvec.ctypes.data_as(dp), tvec.ctypes.data_as(dp), inl.ctypes.data_as(ip)
    fn = lib.ph_solvepnp_ransac_epnp
    t0 = time.perf_counter()
    for r in range(reps):
        o, i = args[r % len(args)]
        fn(o, i, n, Kp, iters, 8.0, 0.99, 0xFFFFFFFFFFFFFFFF, rp, tp, lp, C.byref(n_inl), C.byref(found),
           C.byref(it))
    return (time.perf_counter() - t0) / reps, it.value, n_inl.value


through_python(20)
through_c_abi(20)
py = min(through_python(200) for _ in range(3))
cabi, it, k = min(through_c_abi(200) for _ in range(3))
print(json.dumps({"workload": f"solvePnPRansac, {n} pairs, iterationsCount {iters}, one call per frame",
                  "python_mirror_ms": py * 1e3, "c_abi_ms": cabi * 1e3,
                  "iterations_run_last": it, "inliers_last": k}))
