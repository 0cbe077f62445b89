#!/usr/bin/env python3
"""One-GPU measurements of the five BASELINE.json configurations (C1..C5), each with a
bounded CPU (oracle) figure beside it.  Prints one JSON line per configuration.

C1  500 x 500 cross-check match + RANSAC-EPnP on one frame (host entry points, latency)
C2  10k x 10k kNN k=2 + ratio (device buffers)            -- the bench.py workload
C3  4096 EPnP hypotheses x 2000 correspondences
C4  batched frames x 2000 descriptors vs a 50k-row DB (one call, frames concatenated)
C5  1M queries x 10M train rows (the per-GPU work of the 8-GPU train-sharded run is 1/8 of it)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import datagen  # noqa: E402
import oracle_bind as orc  # noqa: E402
import poseestimation_amd as pa  # noqa: E402
from poseestimation_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1024, help="C4 frames (4096 in BASELINE)")
ap.add_argument("--c5-train", type=int, default=10_000_000)
ap.add_argument("--c5-query", type=int, default=1_000_000)
ap.add_argument("--skip", default="")
args = ap.parse_args()
_lib.init(0)
lib = _lib.load()
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
sp = C.c_void_p(stream.cuda_stream)
THREADS = min(16, len(os.sched_getaffinity(0)))


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def gpu_rand_descriptors(n, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    return torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)


def knn_dev(d_q, d_t, k, d_idx, d_dist, d_keep=None, ratio=0.0):
    _lib.check(lib.ph_knn_hamming256_dev(d_q.data_ptr(), d_q.shape[0], d_t.data_ptr(), d_t.shape[0], k,
                                         0, d_idx.data_ptr(), d_dist.data_ptr(),
                                         d_keep.data_ptr() if d_keep is not None else None, ratio, sp))


out = []
if "C1" not in args.skip:
    q = datagen.random_descriptors(500, 1)
    t = datagen.perturbed_copy(q, 2)
    bf = pa.BFMatcher(pa.NORM_HAMMING, crossCheck=True)
    obj, img, R, tv, inl = datagen.pnp_scene(n=500, outlier_frac=0.3, seed=3)
    m_s = timed(lambda: bf.match(q, t), 30)
    # the same call at the C ABI (no Python DMatch objects)
    u8 = C.POINTER(C.c_uint8)
    raw_out = (_lib.DMatchC * 500)(); raw_n = C.c_int(0)
    mc_s = timed(lambda: lib.ph_match_crosscheck_hamming256(q.ctypes.data_as(u8), 500, 32, t.ctypes.data_as(u8), 500, 32, raw_out, C.byref(raw_n)), 200)
    p_s = timed(lambda: pa.solvePnPRansac(obj, img, datagen.CAMERA_K, None, iterationsCount=100), 10)
    c0 = time.perf_counter(); [orc.match(q, t, cross_check=True) for _ in range(20)]; cm = (time.perf_counter() - c0) / 20
    c0 = time.perf_counter(); [orc.solvepnp_ransac(obj, img, datagen.CAMERA_K, iterations=100) for _ in range(20)]; cp = (time.perf_counter() - c0) / 20
    # the same work for 256 frames at once: one batched cross-check launch (device buffers)
    # and one batched solvePnPRansac call (host buffers)
    from poseestimation_amd import distributed as D
    FB = 256
    d_qb, d_tb = gpu_rand_descriptors(FB * 500, 11), gpu_rand_descriptors(FB * 500, 12)
    offs = np.arange(FB + 1) * 500
    mb_s = timed(lambda: D.match_crosscheck_frames(d_qb, offs, d_tb, offs), 10)
    scenes = [datagen.pnp_scene(n=500, outlier_frac=0.3, seed=300 + f) for f in range(FB)]
    fo, fi = [x[0] for x in scenes], [x[1] for x in scenes]
    pb_s = timed(lambda: pa.solvePnPRansacBatch(fo, fi, datagen.CAMERA_K, None, iterationsCount=100), 5)
    out.append({"config": "C1 500x500 crossCheck match + solvePnPRansac(100 iters, 500 pairs), host buffers, latency",
                "gpu_match_us": m_s * 1e6, "gpu_match_c_abi_us": mc_s * 1e6, "gpu_pnp_us": p_s * 1e6, "cpu_match_us": cm * 1e6, "cpu_pnp_us": cp * 1e6,
                "gpu_256_frames_batched": {"match_us_per_frame": mb_s / FB * 1e6, "pnp_us_per_frame": pb_s / FB * 1e6},
                "cpu": "oracle, 1 thread", "note": "gpu_match_us goes through the Python mirror (500 DMatch objects built per call), gpu_match_c_abi_us is the library call itself; the CPU's sequential RANSAC stops after ~25 iterations"})

if "C2" not in args.skip:
    from poseestimation_amd import distributed as D2
    d_q, d_t = gpu_rand_descriptors(10000, 2), gpu_rand_descriptors(10000, 102)
    d_i = torch.empty((10000, 2), dtype=torch.int32, device=dev); d_d = torch.empty_like(d_i)
    d_k = torch.empty(10000, dtype=torch.uint8, device=dev)
    s = timed(lambda: [knn_dev(d_q, d_t, 2, d_i, d_d, d_k, 0.75) for _ in range(20)], 10) / 20
    qh, th = d_q.cpu().numpy(), d_t.cpu().numpy()
    c0 = time.perf_counter(); ridx, rdist = orc.knn(qh, th, 2, threads=THREADS); cs = time.perf_counter() - c0
    d_q16, d_t16 = gpu_rand_descriptors(160000, 21), gpu_rand_descriptors(160000, 22)
    o16 = np.arange(17) * 10000
    s16 = timed(lambda: D2.knn_frames(d_q16, o16, d_t16, o16, 2, ratio=0.75), 10)
    out.append({"config": "C2 10k x 10k kNN k=2 + ratio, one call", "gpu_us": s * 1e6, "gpu_evals_per_s": 1e8 / s,
                "gpu_16_frames_one_launch_evals_per_s": 16e8 / s16,
                "cpu_evals_per_s": 1e8 / cs, "cpu": f"oracle, {THREADS} threads",
                "bit_exact": bool(np.array_equal(d_i.cpu().numpy(), ridx) and np.array_equal(d_d.cpu().numpy(), rdist))})

if "C3" not in args.skip:
    obj, img, R, tv, inl = datagen.pnp_scene(n=2000, outlier_frac=0.3, seed=3)
    sub = orc.ransac_subsets(2000, 4096)
    d_obj, d_img, d_sub = (torch.from_numpy(x).to(dev) for x in (obj, img, sub))
    d_m = torch.empty((4096, 6), dtype=torch.float64, device=dev); d_c = torch.empty(4096, dtype=torch.int32, device=dev)
    K = np.ascontiguousarray(datagen.CAMERA_K).reshape(9); Kp = K.ctypes.data_as(C.POINTER(C.c_double))
    s = timed(lambda: _lib.check(lib.ph_ransac_epnp_hypotheses_dev(d_obj.data_ptr(), d_img.data_ptr(), 2000, Kp, d_sub.data_ptr(), 4096, 8.0, d_m.data_ptr(), d_c.data_ptr(), sp)), 10)
    c0 = time.perf_counter(); rm, rc = orc.pnp_hypotheses(obj, img, datagen.CAMERA_K, sub[:2048], threads=THREADS); cs = time.perf_counter() - c0
    full = timed(lambda: pa.solvePnPRansac(obj, img, datagen.CAMERA_K, None, iterationsCount=4096), 5)
    out.append({"config": "C3 4096 EPnP hypotheses x 2000 correspondences, 30% outliers, one problem", "gpu_ms": s * 1e3,
                "gpu_hyps_per_s": 4096 / s, "cpu_hyps_per_s": 2048 / cs, "cpu": f"oracle, {THREADS} threads",
                "counts_equal": bool(np.array_equal(d_c.cpu().numpy()[:2048], rc)),
                "solvePnPRansac_4096_host_call_ms": full * 1e3})

if "C4" not in args.skip:
    F = args.frames
    d_t = gpu_rand_descriptors(50000, 4)
    d_q = gpu_rand_descriptors(F * 2000, 40)
    d_i = torch.empty((F * 2000, 2), dtype=torch.int32, device=dev); d_d = torch.empty_like(d_i)
    d_k = torch.empty(F * 2000, dtype=torch.uint8, device=dev)
    s = timed(lambda: knn_dev(d_q, d_t, 2, d_i, d_d, d_k, 0.75), 3)
    rows = np.random.default_rng(0).choice(F * 2000, 512, replace=False)
    qh, th = d_q[torch.from_numpy(rows).to(dev)].cpu().numpy(), d_t.cpu().numpy()
    c0 = time.perf_counter(); ridx, rdist = orc.knn(qh, th, 2, threads=THREADS); cs = time.perf_counter() - c0
    ok = np.array_equal(d_i.cpu().numpy()[rows], ridx) and np.array_equal(d_d.cpu().numpy()[rows], rdist)
    # one frame per call, for comparison
    s1 = timed(lambda: [knn_dev(d_q[f * 2000:(f + 1) * 2000], d_t, 2, d_i[f * 2000:(f + 1) * 2000], d_d[f * 2000:(f + 1) * 2000]) for f in range(32)], 3) / 32
    out.append({"config": f"C4 {F} frames x 2000 descriptors vs 50k-row DB, k=2 + ratio, ONE call (frames concatenated); BASELINE has 4096 frames over 8 GPUs = 512 per GPU",
                "gpu_ms": s * 1e3, "gpu_evals_per_s": F * 2000 * 50000 / s, "one_frame_per_call_us": s1 * 1e6,
                "one_frame_per_call_evals_per_s": 2000 * 50000 / s1,
                "cpu_evals_per_s": 512 * 50000 / cs, "cpu": f"oracle, {THREADS} threads, 512 sampled query rows", "sample_bit_exact": bool(ok)})

if "C5" not in args.skip:
    nq, nt = args.c5_query, args.c5_train
    d_q = gpu_rand_descriptors(nq, 5)
    d_t = gpu_rand_descriptors(nt, 50)
    d_i = torch.empty((nq, 2), dtype=torch.int32, device=dev); d_d = torch.empty_like(d_i)
    t0 = time.perf_counter(); knn_dev(d_q, d_t, 2, d_i, d_d); torch.cuda.synchronize(); s = time.perf_counter() - t0
    t0 = time.perf_counter(); knn_dev(d_q, d_t, 2, d_i, d_d); torch.cuda.synchronize(); s = min(s, time.perf_counter() - t0)
    rows = np.random.default_rng(0).choice(nq, 64, replace=False)
    qh, th = d_q[torch.from_numpy(rows).to(dev)].cpu().numpy(), d_t.cpu().numpy()
    c0 = time.perf_counter(); ridx, rdist = orc.knn(qh, th, 2, threads=THREADS); cs = time.perf_counter() - c0
    ok = np.array_equal(d_i.cpu().numpy()[rows], ridx) and np.array_equal(d_d.cpu().numpy()[rows], rdist)
    # the 8-way train-sharded form on this one GPU: 8 shard searches with global indices + merge
    from poseestimation_amd.distributed import block_bounds
    ci = torch.empty((8, nq, 2), dtype=torch.int32, device=dev); cd = torch.empty_like(ci)
    def sharded():
        for r in range(8):
            lo, hi = block_bounds(nt, 8, r)
            _lib.check(lib.ph_knn_hamming256_dev(d_q.data_ptr(), nq, d_t[lo:hi].data_ptr(), hi - lo, 2, lo, ci[r].data_ptr(), cd[r].data_ptr(), None, 0.0, sp))
    t0 = time.perf_counter(); sharded(); torch.cuda.synchronize(); ss = time.perf_counter() - t0
    d_i2 = torch.empty_like(d_i); d_d2 = torch.empty_like(d_d)
    t0 = time.perf_counter()
    _lib.check(lib.ph_knn_merge_dev(ci.data_ptr(), cd.data_ptr(), 8, nq, 2, d_i2.data_ptr(), d_d2.data_ptr(), None, 0.0, sp))
    torch.cuda.synchronize(); sm = time.perf_counter() - t0
    out.append({"config": f"C5 {nq} queries x {nt} train rows, k=2, one GPU", "gpu_s": s, "gpu_evals_per_s": nq * nt / s,
                "cpu_evals_per_s": 64 * nt / cs, "cpu": f"oracle, {THREADS} threads, 64 sampled query rows", "sample_bit_exact": bool(ok),
                "as_8_shards_s": ss, "shard_merge_ms": sm * 1e3, "per_shard_s": ss / 8,
                "sharded_equals_unsharded": bool(torch.equal(d_i, d_i2) and torch.equal(d_d, d_d2)),
                "allgather_bytes_per_rank": nq * 2 * 4 * 2})

for o in out:
    print(json.dumps(o), flush=True)
