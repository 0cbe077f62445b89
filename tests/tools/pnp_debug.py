import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import datagen, oracle_bind as orc
from poseestimation_amd import _lib
from test_pnp_gpu import gpu_hypotheses
_lib.init(0)
obj, img, R, t, inl = datagen.pnp_scene(n=2000, seed=3)
subsets = orc.ransac_subsets(2000, 512)
models, counts = gpu_hypotheses(_lib, obj, img, datagen.CAMERA_K, subsets)
rmodels, rcounts = orc.pnp_hypotheses(obj, img, datagen.CAMERA_K, subsets)
d = np.abs(models - rmodels).max(axis=1)
print("count equal frac", (counts == rcounts).mean())
print("model maxabs diff quantiles", np.quantile(d, [0, .1, .25, .5, .75, .9, 1]))
order = np.argsort(-rcounts)
for h in list(order[:8]) + list(order[200:206]) + list(order[-6:]):
    print(h, "cpu", rcounts[h], "gpu", counts[h], "dmodel %.3e" % d[h], "true-inlier sample" , inl[subsets[h]].all())
# exact data single solves
for n in (5, 6, 50):
    o, i, R, t, _ = datagen.pnp_scene(n=n, outlier_frac=0.0, noise_px=0.0, seed=2)
    import poseestimation_amd as pa
    ok, r, tv = pa.solvePnP(o, i, datagen.CAMERA_K)
    r0, t0 = orc.solvepnp_epnp(o, i, datagen.CAMERA_K)
    print("n", n, "gpu-cpu", np.abs(r.ravel() - r0).max(), np.abs(tv.ravel() - t0).max(), "gpu-gt", np.abs(orc.rodrigues(r.ravel()) - R).max())
