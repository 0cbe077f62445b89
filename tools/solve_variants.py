"""A/B of the EPnP solve variants on the benchmark's RANSAC step (16 problems x 4096 hypotheses
x 2000 correspondences in one batched call): single kernel (mode 1), two kernels (mode 2: the
decomposition at two waves per SIMD, then the rest at one), two kernels with the second half at
two waves per SIMD too (mode 3).  Prints the step time, the solve time of one round of 32768
hypotheses (HIP events inside the library) and whether models and counts are bit-identical to
mode 1.   python tools/solve_variants.py"""
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
P, H, n = 16, 4096, 2000
K = np.ascontiguousarray(datagen.CAMERA_K).reshape(9)
Kp = K.ctypes.data_as(C.POINTER(C.c_double))
objs, imgs, subs = [], [], []
for p in range(P):
    obj, img, R, t, inl = datagen.pnp_scene(n=n, outlier_frac=0.3, seed=3 + p)
    sub = np.empty((H, 5), np.int32)
    pa._lib.check(lib.ph_ransac_subsets(0xFFFFFFFFFFFFFFFF, n, H, sub.ctypes.data_as(C.POINTER(C.c_int32))))
    objs.append(obj); imgs.append(img); subs.append(sub)
d_obj = torch.from_numpy(np.concatenate(objs)).to(dev)
d_img = torch.from_numpy(np.concatenate(imgs)).to(dev)
d_sub = torch.from_numpy(np.concatenate(subs)).to(dev)
d_off = torch.arange(0, (P + 1) * n, n, dtype=torch.int32, device=dev)
stream = torch.cuda.Stream()
sptr = C.c_void_p(stream.cuda_stream)
ref = None
for mode in (1, 2, 3, 1, 2, 3):
    pa._lib.check(lib.ph_ransac_set_solve_path(mode))
    d_m = torch.zeros((P * H, 6), dtype=torch.float64, device=dev)
    d_c = torch.zeros(P * H, dtype=torch.int32, device=dev)

    def step():
        pa._lib.check(lib.ph_ransac_epnp_hypotheses_batched_dev(
            d_obj.data_ptr(), d_img.data_ptr(), d_off.data_ptr(), P, Kp, d_sub.data_ptr(), H, 8.0,
            d_m.data_ptr(), d_c.data_ptr(), sptr))
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 10
    pa._lib.check(lib.ph_set_profiling(1))
    a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
    first, last = [], []
    for _ in range(10):
        step(); step()
        pa._lib.check(lib.ph_ransac_last_timing(C.byref(a), C.byref(b), C.byref(c)))
        first.append(a.value); last.append(b.value)
    pa._lib.check(lib.ph_set_profiling(0))
    m, cnt = d_m.cpu().numpy(), d_c.cpu().numpy()
    if ref is None:
        ref = (m.copy(), cnt.copy())
    same = bool(np.array_equal(m.view(np.uint64), ref[0].view(np.uint64)) and np.array_equal(cnt, ref[1]))
    print(json.dumps({"solve_path": mode, "ms_per_step": ms, "hyps_per_s": P * H / (ms * 1e-3),
                      "first_round_solve_ms": float(np.mean(first)), "last_round_solve_ms": float(np.mean(last)),
                      "identical_to_single_kernel": same}), flush=True)
pa._lib.check(lib.ph_ransac_set_solve_path(0))
