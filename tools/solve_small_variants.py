"""Latency of ONE problem of a few thousand hypotheses (BASELINE configs[2] as a single call:
4096 hypotheses x 2000 correspondences), solve + score, per solve path.  It served as the A/B
that moved 2049..4096 hypotheses from 8 to 4 per wave (the run is kept as
profiles/r02/solve_small_4_per_wave.jsonl); both paths it names now pick the same kernel.
python tools/solve_small_variants.py"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import datagen
import poseestimation_amd as pa

pa._lib.init(0)
lib = pa._lib.load()
dev = torch.device("cuda", 0)
n = 2000
K = np.ascontiguousarray(datagen.CAMERA_K).reshape(9)
Kp = K.ctypes.data_as(C.POINTER(C.c_double))
obj, img, R, t, inl = datagen.pnp_scene(n=n, outlier_frac=0.3, seed=3)
d_obj, d_img = torch.from_numpy(obj).to(dev), torch.from_numpy(img).to(dev)
stream = torch.cuda.Stream()
sptr = C.c_void_p(stream.cuda_stream)
for H in (2049, 3072, 4096, 6000, 8192):
    sub = np.empty((H, 5), np.int32)
    pa._lib.check(lib.ph_ransac_subsets(0xFFFFFFFFFFFFFFFF, n, H, sub.ctypes.data_as(C.POINTER(C.c_int32))))
    d_sub = torch.from_numpy(sub).to(dev)
    ref = None
    for mode in (0, 1, 0, 1):
        pa._lib.check(lib.ph_ransac_set_solve_path(mode))
        d_m = torch.zeros((H, 6), dtype=torch.float64, device=dev)
        d_c = torch.zeros(H, dtype=torch.int32, device=dev)

        def call():
            pa._lib.check(lib.ph_ransac_epnp_hypotheses_dev(d_obj.data_ptr(), d_img.data_ptr(), n, Kp,
                                                            d_sub.data_ptr(), H, 8.0, d_m.data_ptr(),
                                                            d_c.data_ptr(), sptr))
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            call()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 50 * 1e3
        m, cnt = d_m.cpu().numpy(), d_c.cpu().numpy()
        if ref is None:
            ref = (m.copy(), cnt.copy())
        same = bool(np.array_equal(m.view(np.uint64), ref[0].view(np.uint64)) and np.array_equal(cnt, ref[1]))
        print(json.dumps({"hypotheses": H, "solve_path": mode, "ms_per_call": ms,
                          "identical_to_default": same}), flush=True)
pa._lib.check(lib.ph_ransac_set_solve_path(0))
