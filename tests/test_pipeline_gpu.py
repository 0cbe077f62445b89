"""GPU suite: the whole hot path for a batch of frames, as the reference's loop would run it
(src/cereal_box.cpp:262-295 per rendered frame): descriptors of every frame matched against
the reference frame's descriptors (k = 2 + ratio test), reference keypoints lifted to the box,
one batched solvePnPRansac, and the GL model matrix of every estimated pose.  Every stage is
compared with the CPU oracle run frame by frame."""
import numpy as np
import pytest

import datagen
import oracle_bind as orc

pytestmark = pytest.mark.gpu


def _flip_bits(desc, rng, max_flips):
    bits = np.unpackbits(desc, axis=1)
    for i in range(bits.shape[0]):
        pos = rng.choice(256, size=int(rng.integers(0, max_flips + 1)), replace=False)
        bits[i, pos] ^= 1
    return np.packbits(bits, axis=1)


def test_frames_to_poses(posehip):
    import torch
    import poseestimation_amd as pa
    from poseestimation_amd import _lib, geometry as g
    rng = np.random.default_rng(21)
    K = g.camera_matrix()
    # reference frame: 500 keypoints on the box, their descriptors and 3-D positions
    M_ref = g.model_matrix(0.8)
    R_ref, t_ref = g.pose_from_model_matrix(M_ref)
    px = rng.random((6000, 2)) * [800, 600]
    pts3d, hit = g.lift_pixels_to_box(px, R_ref, t_ref, K)
    pts3d = pts3d[hit][:500]
    assert len(pts3d) == 500
    ref_desc = datagen.random_descriptors(500, 22)
    # F frames: the box a little further round each time; 400 of the reference keypoints
    # re-detected (noisy pixel, a few flipped descriptor bits) + 100 clutter keypoints
    F = 12
    frames, truth = [], []
    for f in range(F):
        M = g.model_matrix(0.8 + 0.04 * (f + 1))
        R, t = g.pose_from_model_matrix(M)
        seen = rng.permutation(500)[:400]
        pix = g.project_cv(pts3d[seen], R, t, K) + rng.normal(0, 0.5, (400, 2))
        desc = _flip_bits(ref_desc[seen], rng, 25)
        pix = np.concatenate([pix, rng.random((100, 2)) * [800, 600]])
        desc = np.concatenate([desc, datagen.random_descriptors(100, 100 + f)])
        order = rng.permutation(500)
        frames.append((np.ascontiguousarray(desc[order]), pix[order]))
        truth.append(M)
    # stage 1: all frames' descriptors against the reference set in ONE launch
    q_all = np.concatenate([fr[0] for fr in frames])
    dev = torch.device("cuda", 0)
    d_q, d_t = torch.from_numpy(q_all).to(dev), torch.from_numpy(ref_desc).to(dev)
    idx = torch.empty((F * 500, 2), dtype=torch.int32, device=dev)
    dist = torch.empty_like(idx)
    keep = torch.empty(F * 500, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    _lib.check(_lib.load().ph_knn_hamming256_dev(d_q.data_ptr(), F * 500, d_t.data_ptr(), 500, 2, 0,
                                                 idx.data_ptr(), dist.data_ptr(), keep.data_ptr(),
                                                 0.75, None))
    _lib.check(_lib.load().ph_stream_synchronize(None))
    idx, dist, keep = idx.cpu().numpy(), dist.cpu().numpy(), keep.cpu().numpy().astype(bool)
    ridx, rdist = orc.knn(q_all, ref_desc, 2, threads=8)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    kept = {m[0] for m in orc.ratio_filter(ridx, rdist, 0.75)}
    assert set(np.nonzero(keep)[0].tolist()) == kept
    # stage 2: 2D-3D pairs of every frame, one batched solvePnPRansac
    objs, imgs = [], []
    for f in range(F):
        rows = np.nonzero(keep[f * 500:(f + 1) * 500])[0]
        objs.append(pts3d[idx[f * 500 + rows, 0]].astype(np.float32))
        imgs.append(frames[f][1][rows].astype(np.float32))
        assert len(rows) >= 350
    res = pa.solvePnPRansacBatch(objs, imgs, K, None, iterationsCount=100)
    for f, (ok, rvec, tvec, inl) in enumerate(res):
        assert ok and len(inl) >= 0.9 * len(objs[f])
        ok0, r0, t0, inl0, _ = orc.solvepnp_ransac(objs[f], imgs[f], K, iterations=100)
        assert ok0 and np.array_equal(inl.ravel(), inl0)
        Ra, Rb = orc.rodrigues(rvec.ravel()), orc.rodrigues(np.ravel(r0))
        assert np.linalg.norm(Ra - Rb) / np.linalg.norm(Ra) <= 1e-4
        assert np.linalg.norm(tvec.ravel() - np.ravel(t0)) / np.linalg.norm(tvec) <= 1e-4
        # stage 3: the wireframe's model matrix from the estimate (src/cereal_box.cpp:286)
        M_est = g.model_matrix_from_pose(rvec, tvec)
        assert np.abs(M_est - truth[f]).max() < 1e-2


def test_measured_ceilings_are_plausible(posehip):
    """The three live ceilings the benchmark divides by: integer VALU (xor + popcount mix),
    fp64 VALU (fma; mul + add) and the FP4 MFMA of the matrix-core matcher.  Each must lie
    between a third of and slightly above the figure the MI355X data sheet implies; invalid
    arguments are refused."""
    import ctypes as C
    lib = posehip.load()
    ops, fma, mix, mf = C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0)
    posehip.check(lib.ph_valu_popcount_peak(500, C.byref(ops)))
    posehip.check(lib.ph_valu_fp64_peak(500, C.byref(fma), C.byref(mix)))
    posehip.check(lib.ph_mfma_fp4_peak(500, C.byref(mf)))
    spec_int = 256 * 4 * 16 * 2.4e9          # lane-ops/s: 1024 SIMDs x 16 lanes per clock at 2.4 GHz
    # wide bounds: the boxes of a pool differ by 10 % and more in the clock they hold under load
    assert spec_int / 3 <= ops.value <= spec_int * 1.3
    assert 78.6e12 / 2 / 3 <= fma.value <= 78.6e12 / 2 * 1.2      # lane-ops/s; 2 FLOP each
    assert fma.value / 3 <= mix.value <= fma.value * 1.5          # (mul + add can out-issue fma: less power)
    assert 10e15 / 3 <= mf.value <= 10e15 * 1.1
    assert lib.ph_mfma_fp4_peak(0, C.byref(mf)) == 1 and lib.ph_mfma_fp4_peak(10, None) == 1
