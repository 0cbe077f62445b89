"""Times solvePnPRansac over F frames: one batched call against F single calls (host entry points)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import datagen
import poseestimation_amd as pa
pa._lib.init(0)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 100
scenes = [datagen.pnp_scene(n=n, outlier_frac=0.3, seed=200 + f) for f in range(F)]
objs, imgs = [s[0] for s in scenes], [s[1] for s in scenes]
for _ in range(3):
    pa.solvePnPRansacBatch(objs, imgs, datagen.CAMERA_K, None, iterationsCount=iters)
    pa.solvePnPRansac(objs[0], imgs[0], datagen.CAMERA_K, None, iterationsCount=iters)
R = 10
t0 = time.perf_counter()
for _ in range(R):
    res = pa.solvePnPRansacBatch(objs, imgs, datagen.CAMERA_K, None, iterationsCount=iters)
tb = (time.perf_counter() - t0) / R
t0 = time.perf_counter()
for _ in range(R):
    single = [pa.solvePnPRansac(o, i, datagen.CAMERA_K, None, iterationsCount=iters) for o, i in zip(objs, imgs)]
ts = (time.perf_counter() - t0) / R
same = all(np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3]) for a, b in zip(res, single))
print({"frames": F, "points": n, "iterations": iters, "batched_ms": tb * 1e3, "batched_us_per_frame": tb / F * 1e6,
       "single_calls_ms": ts * 1e3, "single_us_per_frame": ts / F * 1e6, "poses_per_s_batched": F / tb, "identical": bool(same)})
