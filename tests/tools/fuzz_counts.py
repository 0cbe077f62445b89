"""Randomised differential run of the hypothesis entry points (solve + scoring) against the CPU
oracle: EVERY hypothesis' inlier count must equal the oracle's and every finite model agree to
1e-11, over random scenes, correspondence counts, thresholds, outlier fractions and batch shapes
(single problems and ragged batches).  Not part of the test suite:
    python tests/tools/fuzz_counts.py [seconds] [seed]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import datagen
import oracle_bind as orc
import poseestimation_amd as pa

pa._lib.init(0)
lib = pa._lib.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
Kc = np.ascontiguousarray(datagen.CAMERA_K, np.float64).reshape(9)
Kp = Kc.ctypes.data_as(C.POINTER(C.c_double))
t_end, last = time.time() + budget, time.time()
n_hyp_total, n_calls, worst = 0, 0, 0.0
while time.time() < t_end:
    if time.time() - last > 20:
        print({"progress_hypotheses": n_hyp_total}, flush=True)
        last = time.time()
    P = int(rng.choice([1, 1, 2, 3, 7]))
    H = int(rng.choice([1, 2, 3, 31, 64, 257, 1024]))
    thr = float(rng.choice([0.5, 1.0, 3.0, 8.0, 20.0]))
    sizes = [int(rng.choice([5, 6, 63, 64, 65, 200, 1000, 2000, 3001])) for _ in range(P)]
    scenes = [datagen.pnp_scene(n=n, outlier_frac=float(rng.choice([0.0, 0.3, 0.6, 1.0])),
                                noise_px=float(rng.choice([0.0, 1.0, 4.0])), seed=int(rng.integers(1 << 30)),
                                angle_deg=float(rng.uniform(-85, 85)), on_faces=bool(rng.random() < 0.3),
                                t=(float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), float(rng.uniform(2.5, 9))))
              for n in sizes]
    subs = [orc.ransac_subsets(n, H, rng_state=int(rng.integers(1, 1 << 62))) for n in sizes]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    d_obj = torch.from_numpy(np.concatenate([s[0] for s in scenes])).to(dev)
    d_img = torch.from_numpy(np.concatenate([s[1] for s in scenes])).to(dev)
    d_sub = torch.from_numpy(np.concatenate(subs)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_m = torch.zeros((P * H, 6), dtype=torch.float64, device=dev)
    d_c = torch.full((P * H,), -7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    if P == 1 and rng.random() < 0.5:
        pa._lib.check(lib.ph_ransac_epnp_hypotheses_dev(d_obj.data_ptr(), d_img.data_ptr(), sizes[0], Kp,
                                                        d_sub.data_ptr(), H, thr, d_m.data_ptr(),
                                                        d_c.data_ptr(), None))
    else:
        pa._lib.check(lib.ph_ransac_epnp_hypotheses_batched_dev(d_obj.data_ptr(), d_img.data_ptr(),
                                                                d_off.data_ptr(), P, Kp, d_sub.data_ptr(), H,
                                                                thr, d_m.data_ptr(), d_c.data_ptr(), None))
    pa._lib.check(lib.ph_stream_synchronize(None))
    m, c = d_m.cpu().numpy(), d_c.cpu().numpy()
    for p in range(P):
        rm, rc = orc.pnp_hypotheses(scenes[p][0], scenes[p][1], datagen.CAMERA_K, subs[p], threshold=thr, threads=16)
        gm, gc = m[p * H:(p + 1) * H], c[p * H:(p + 1) * H]
        if not np.array_equal(gc, rc) and os.environ.get("FUZZ_DUMP"):
            np.savez(os.environ["FUZZ_DUMP"], obj=scenes[p][0], img=scenes[p][1], sub=subs[p], thr=thr,
                     gm=gm, gc=gc, rm=rm, rc=rc)
        assert np.array_equal(gc, rc), (sizes[p], H, thr, np.nonzero(gc != rc)[0][:5], gc[gc != rc][:5], rc[gc != rc][:5])
        fin = np.isfinite(rm)
        assert np.array_equal(np.isfinite(gm), fin)
        if fin.any():
            d = float(np.abs(gm[fin] - rm[fin]).max())
            worst = max(worst, d)
            assert d <= 1e-11, (sizes[p], H, d)
    n_hyp_total += P * H
    n_calls += 1
print({"calls": n_calls, "hypotheses": n_hyp_total, "worst_abs_model_difference": worst, "ok": True})
