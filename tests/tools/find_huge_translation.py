"""Searches random 5-point samples of the fixture's scene for a hypothesis whose solve puts the
camera more than 1e37 units away (the denormal-reciprocal case of the scoring kernel's first
pass) and rewrites tests/golden/ransac_huge_translation.npz with 8 samples around it and the
ORACLE's models and counts for them.  Run on the GPU box; the solve runs on the GPU only to
find candidates, what is stored comes from the oracle."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import datagen
import oracle_bind as orc
from poseestimation_amd import _lib

_lib.init(0)
lib = _lib.load()
dev = torch.device("cuda", 0)
path = os.path.join(ROOT, "tests", "golden", "ransac_huge_translation.npz")
g = np.load(path)
obj, img, thr = g["obj"], g["img"], float(g["thr"])
n = obj.shape[0]
Kc = np.ascontiguousarray(datagen.CAMERA_K, np.float64).reshape(9)
Kp = Kc.ctypes.data_as(C.POINTER(C.c_double))
d_obj, d_img = torch.from_numpy(obj).to(dev), torch.from_numpy(img).to(dev)
H = 1 << 22
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
found = None
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    sub = orc.ransac_subsets(n, H, rng_state=int(rng.integers(1, 1 << 62)))
    d_sub = torch.from_numpy(sub).to(dev)
    d_m = torch.zeros((H, 6), dtype=torch.float64, device=dev)
    d_c = torch.zeros(H, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    _lib.check(lib.ph_ransac_epnp_hypotheses_dev(d_obj.data_ptr(), d_img.data_ptr(), n, Kp, d_sub.data_ptr(), H, thr,
                                                 d_m.data_ptr(), d_c.data_ptr(), None))
    _lib.check(lib.ph_stream_synchronize(None))
    t = d_m[:, 3:].abs()
    t = torch.where(torch.isfinite(t), t, torch.zeros_like(t)).max(dim=1).values
    t = torch.where(t < 3.0e38, t, torch.zeros_like(t))       # the window in which 1 / t is an fp32 denormal
    best = int(torch.argmax(t).item())
    print({"batch": it, "largest_translation_below_3e38": float(t[best].item())}, flush=True)
    if float(t[best].item()) > 1e37:
        lo = max(0, min(best - 3, H - 8))
        found = sub[lo:lo + 8].copy()
        break
if found is None:
    print("nothing found")
    sys.exit(1)
models, counts = orc.pnp_hypotheses(obj, img, datagen.CAMERA_K, found, threshold=thr)
big = np.abs(models[:, 3:]).max(axis=1)
assert ((big > 1e37) & (big < 3.0e38)).any()
np.savez(os.path.join(ROOT, "gpurun_out", "ransac_huge_translation.npz"), obj=obj, img=img, sub=found, thr=np.float64(thr),
         counts=counts, models=models)
print({"rows_with_huge_translation": np.nonzero((big > 1e37) & (big < 3.0e38))[0].tolist(), "counts": counts.tolist()})
