"""Randomised differential run of solvePnPRansac (batched and single) against the CPU oracle:
random correspondence counts, outlier fractions, noise, iteration budgets, thresholds.
Not part of the test suite:  python tests/tools/fuzz_pnp.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import datagen, oracle_bind as orc
import poseestimation_amd as pa
pa._lib.init(0)
import ctypes as C
_lib = pa._lib
_K9 = np.ascontiguousarray(datagen.CAMERA_K, np.float64).reshape(9)


def single_call(obj, img, iters, thr):
    """ph_solvepnp_ransac_epnp on one problem (the lone-call path, which chooses the model on
    the GPU): (found, rvec, tvec, inliers, iterations_run)."""
    fp, dp, ip = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int32)
    n = obj.shape[0]
    rvec, tvec, inl = np.zeros(3), np.zeros(3), np.empty(n, np.int32)
    n_inl, found, it = C.c_int(0), C.c_int(0), C.c_int(0)
    _lib.check(_lib.load().ph_solvepnp_ransac_epnp(
        obj.ctypes.data_as(fp), img.ctypes.data_as(fp), n, _K9.ctypes.data_as(dp), iters, thr, 0.99,
        0xFFFFFFFFFFFFFFFF, rvec.ctypes.data_as(dp), tvec.ctypes.data_as(dp), inl.ctypes.data_as(ip),
        C.byref(n_inl), C.byref(found), C.byref(it)))
    return bool(found.value), rvec, tvec, inl[:n_inl.value], it.value


n_single = 0
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end, n_prob, worst = time.time() + budget, 0, 0.0
last = time.time()
while time.time() < t_end:
    if time.time() - last > 20:
        print({'progress': n_prob}, flush=True)
        last = time.time()
    P = int(rng.integers(1, 24))
    iters = int(rng.choice([1, 7, 50, 100, 333]))
    thr = float(rng.choice([1.0, 3.0, 8.0]))
    scenes = []
    for _ in range(P):
        n = int(rng.choice([5, 6, 7, 20, 64, 65, 200, 500, 900]))
        scenes.append(datagen.pnp_scene(n=n, outlier_frac=float(rng.choice([0.0, 0.2, 0.5, 0.9])),
                                        noise_px=float(rng.choice([0.0, 0.5, 2.0])), seed=int(rng.integers(1 << 30)),
                                        angle_deg=float(rng.uniform(-80, 80)), on_faces=bool(rng.random() < 0.3)))
    objs, imgs = [s[0] for s in scenes], [s[1] for s in scenes]
    res = pa.solvePnPRansacBatch(objs, imgs, datagen.CAMERA_K, None, iterationsCount=iters,
                                 reprojectionError=thr, returnIterations=True)
    for obj, img, (ok, rvec, tvec, inl, it) in zip(objs, imgs, res):
        ok0, r0, t0, inl0, it0 = orc.solvepnp_ransac(obj, img, datagen.CAMERA_K, iterations=iters, reproj_err=thr)
        assert ok == ok0, (len(obj), iters, thr, ok, ok0)
        if len(obj) > 5:
            assert it == it0, (len(obj), iters, it, it0)
        if ok:
            assert np.array_equal(inl.ravel(), inl0), (len(obj), iters, thr)
            Ra, Rb = orc.rodrigues(rvec.ravel()), orc.rodrigues(np.ravel(r0))
            e = max(np.linalg.norm(Ra - Rb) / np.linalg.norm(Ra),
                    np.linalg.norm(tvec.ravel() - np.ravel(t0)) / np.linalg.norm(tvec))
            if np.isfinite(e):
                worst = max(worst, e)
                assert e <= 1e-4, (len(obj), iters, thr, e)
        n_prob += 1
    # the same problems, one call each (every third batch): equal to the batched call's result
    if n_prob % 3 == 0:
        for obj, img, (ok, rvec, tvec, inl, it) in zip(objs, imgs, res):
            if len(obj) < 5:
                continue
            ok1, r1, t1, inl1, it1 = single_call(obj, img, iters, thr)
            assert ok1 == ok, (len(obj), iters, thr, ok1, ok)
            if len(obj) > 5:
                assert it1 == it, (len(obj), iters, thr, it1, it)
            if ok:
                assert np.array_equal(inl1, inl.ravel()), (len(obj), iters, thr)
                assert np.abs(r1 - rvec.ravel()).max() <= 1e-12 and np.abs(t1 - tvec.ravel()).max() <= 1e-12
            n_single += 1
print({"problems": n_prob, "single_calls": n_single, "worst_relative_pose_difference": worst, "ok": True})
