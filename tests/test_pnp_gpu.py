"""GPU suite: RANSAC-EPnP on the GPU against the CPU oracle.

Tolerances: poses must agree within 1e-4 relative rotation / translation error
(the metric of the reference's src/evaluation/Evaluation.py:5-15) for identical
RANSAC seeds; inlier counts and index lists are compared exactly.
"""
import ctypes as C

import numpy as np
import pytest

import datagen
import oracle_bind as orc

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-4


def rel_translation_error(t_est, t_gt):
    return np.linalg.norm(t_gt - t_est) / np.linalg.norm(t_est)


def rel_rotation_error(R_est, R_gt):
    return np.linalg.norm(R_gt - R_est) / np.linalg.norm(R_est)


def assert_same_pose(r_a, t_a, r_b, t_b, tol=POSE_TOL):
    Ra, Rb = orc.rodrigues(np.ravel(r_a)), orc.rodrigues(np.ravel(r_b))
    assert rel_rotation_error(Ra, Rb) <= tol
    assert rel_translation_error(np.ravel(t_a), np.ravel(t_b)) <= tol


def gpu_hypotheses(lib, obj, img, K, subsets, thr=8.0):
    import torch
    dev = torch.device("cuda", 0)
    d_obj = torch.from_numpy(obj).to(dev)
    d_img = torch.from_numpy(img).to(dev)
    d_sub = torch.from_numpy(subsets).to(dev)
    h = subsets.shape[0]
    d_models = torch.zeros((h, 6), dtype=torch.float64, device=dev)
    d_counts = torch.zeros(h, dtype=torch.int32, device=dev)
    Kc = np.ascontiguousarray(K, np.float64).reshape(9)
    torch.cuda.synchronize()
    lib.check(lib.load().ph_ransac_epnp_hypotheses_dev(
        d_obj.data_ptr(), d_img.data_ptr(), obj.shape[0], Kc.ctypes.data_as(C.POINTER(C.c_double)),
        d_sub.data_ptr(), h, thr, d_models.data_ptr(), d_counts.data_ptr(), None))
    lib.check(lib.load().ph_stream_synchronize(None))
    return d_models.cpu().numpy(), d_counts.cpu().numpy()


def test_subset_stream_matches_cv_rng(posehip):
    want = orc.ransac_subsets(2000, 300)
    got = np.empty((300, 5), np.int32)
    posehip.check(posehip.load().ph_ransac_subsets(orc.CV_RNG_DEFAULT, 2000, 300,
                                                   got.ctypes.data_as(C.POINTER(C.c_int32))))
    assert np.array_equal(got, want)
    assert all(len(set(r)) == 5 for r in got.tolist())


def assert_models_equal(models, rmodels, tol=1e-12):
    """Hypothesis models against the oracle's: NaNs (degenerate samples) in the same places,
    finite entries within tol absolute (rvec in radians, tvec in model units)."""
    assert np.array_equal(np.isnan(models), np.isnan(rmodels))
    fin = np.isfinite(rmodels)
    assert np.array_equal(np.isfinite(models), fin)
    assert np.abs(models[fin] - rmodels[fin]).max() <= tol


@pytest.mark.parametrize("seed,on_faces", [(3, False), (4, True)])
def test_hypotheses_match_oracle(posehip, seed, on_faces):
    obj, img, R, t, inl = datagen.pnp_scene(n=2000, seed=seed, on_faces=on_faces)
    subsets = orc.ransac_subsets(2000, 512)
    models, counts = gpu_hypotheses(posehip, obj, img, datagen.CAMERA_K, subsets)
    rmodels, rcounts = orc.pnp_hypotheses(obj, img, datagen.CAMERA_K, subsets, threads=8)
    assert np.array_equal(counts, rcounts)          # every inlier count, exactly
    assert_models_equal(models, rmodels)
    good = np.nonzero(rcounts >= 0.5 * rcounts.max())[0]
    assert len(good) > 10


def test_full_size_config_hypotheses_exact(posehip):
    """BASELINE configs[2] at full size: 4096 EPnP hypotheses (cv::RNG's first 4096 samples)
    scored on 2000 correspondences with 30 % outliers.  Every inlier count equals the
    oracle's, every model agrees to 1e-12, and the best hypothesis is the scene's pose."""
    obj, img, R, t, inl = datagen.pnp_scene(n=2000, outlier_frac=0.3, seed=3)
    subsets = orc.ransac_subsets(2000, 4096)
    models, counts = gpu_hypotheses(posehip, obj, img, datagen.CAMERA_K, subsets)
    rmodels, rcounts = orc.pnp_hypotheses(obj, img, datagen.CAMERA_K, subsets, threads=8)
    assert np.array_equal(counts, rcounts)
    assert_models_equal(models, rmodels)
    best = int(counts.argmax())
    assert counts[best] >= 0.6 * 2000                      # about 70 % of the rows are inliers
    assert rel_rotation_error(orc.rodrigues(models[best, :3]), R) < 2e-2
    assert rel_translation_error(models[best, 3:], t) < 2e-2


def test_degenerate_sample_with_huge_translation(posehip):
    """A case found by tests/tools/fuzz_counts.py: a degenerate 5-point sample whose EPnP solve puts
    the camera 1e38 units away.  The depth's reciprocal is then a denormal in fp32, which the
    scoring kernel's fp32 first pass must not rely on (it once read it as zero and counted a
    row at the image centre as an inlier).  Inputs and the oracle's outputs are kept in
    tests/golden/ransac_huge_translation.npz (8 samples around the failing one)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ransac_huge_translation.npz"))
    models, counts = gpu_hypotheses(posehip, g["obj"], g["img"], datagen.CAMERA_K, g["sub"],
                                    thr=float(g["thr"]))
    assert np.abs(g["models"][3, 3:]).max() > 1e37          # the sample in question
    rmodels, rcounts = orc.pnp_hypotheses(g["obj"], g["img"], datagen.CAMERA_K, g["sub"],
                                          threshold=float(g["thr"]))
    assert np.array_equal(rcounts, g["counts"]) and np.array_equal(counts, rcounts)
    assert_models_equal(models, rmodels)


def test_epnp_solve_matches_oracle(posehip):
    import poseestimation_amd as pa
    for n, seed in [(4, 1), (5, 2), (6, 3), (50, 4), (700, 5)]:
        obj, img, R, t, _ = datagen.pnp_scene(n=n, outlier_frac=0.0, noise_px=0.5, seed=seed)
        ok, rvec, tvec = pa.solvePnP(obj, img, datagen.CAMERA_K, None, flags=pa.SOLVEPNP_EPNP)
        r0, t0 = orc.solvepnp_epnp(obj, img, datagen.CAMERA_K)
        assert ok
        assert_same_pose(rvec, tvec, r0, t0, 1e-9)


@pytest.mark.parametrize("n,outliers,iters,seed", [(2000, 0.3, 4096, 3), (2000, 0.3, 100, 7),
                                                   (500, 0.5, 1000, 8), (64, 0.1, 100, 9)])
def test_solvepnp_ransac_matches_oracle(posehip, n, outliers, iters, seed):
    import poseestimation_amd as pa
    obj, img, R, t, inl = datagen.pnp_scene(n=n, outlier_frac=outliers, seed=seed)
    ok, rvec, tvec, inliers = pa.solvePnPRansac(obj, img, datagen.CAMERA_K, None,
                                                iterationsCount=iters, reprojectionError=8.0,
                                                confidence=0.99, flags=pa.SOLVEPNP_EPNP)
    ok0, r0, t0, inl0, iters0 = orc.solvepnp_ransac(obj, img, datagen.CAMERA_K, iterations=iters)
    assert ok == ok0 and ok
    assert np.array_equal(inliers.ravel(), inl0)
    assert_same_pose(rvec, tvec, r0, t0)
    # and the pose is the scene's pose (1 px noise -> well within a percent)
    assert rel_rotation_error(orc.rodrigues(rvec.ravel()), R) < 1e-2
    assert rel_translation_error(tvec.ravel(), t) < 1e-2


def test_iterations_run_follows_the_adaptive_budget(posehip):
    obj, img, R, t, inl = datagen.pnp_scene(n=2000, outlier_frac=0.3, seed=3)
    K = np.ascontiguousarray(datagen.CAMERA_K).reshape(9)
    rvec, tvec = np.zeros(3), np.zeros(3)
    inl_out = np.empty(2000, np.int32)
    n_inl, found, iters = C.c_int(0), C.c_int(0), C.c_int(0)
    fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
    posehip.check(posehip.load().ph_solvepnp_ransac_epnp(
        obj.ctypes.data_as(fp), img.ctypes.data_as(fp), 2000, K.ctypes.data_as(dp), 4096, 8.0,
        0.99, orc.CV_RNG_DEFAULT, rvec.ctypes.data_as(dp), tvec.ctypes.data_as(dp),
        inl_out.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n_inl), C.byref(found),
        C.byref(iters)))
    ok0, r0, t0, inl0, iters0 = orc.solvepnp_ransac(obj, img, datagen.CAMERA_K, iterations=4096)
    assert found.value == 1 and iters.value == iters0 and n_inl.value == len(inl0)


def test_five_points_and_all_outliers(posehip):
    import poseestimation_amd as pa
    obj, img, R, t, _ = datagen.pnp_scene(n=5, outlier_frac=0.0, noise_px=0.0, seed=11)
    ok, rvec, tvec, inliers = pa.solvePnPRansac(obj, img, datagen.CAMERA_K, None)
    assert ok and inliers.ravel().tolist() == [0, 1, 2, 3, 4]
    r0, t0 = orc.solvepnp_epnp(obj, img, datagen.CAMERA_K)
    assert_same_pose(rvec, tvec, r0, t0, 1e-9)
    # pure noise: OpenCV's rule still accepts any model with >= 5 inliers at 8 px
    obj, img, R, t, _ = datagen.pnp_scene(n=300, outlier_frac=1.0, seed=12)
    ok, rvec, tvec, inliers = pa.solvePnPRansac(obj, img, datagen.CAMERA_K, None,
                                                iterationsCount=50, reprojectionError=1.0)
    ok0, r0, t0, inl0, _ = orc.solvepnp_ransac(obj, img, datagen.CAMERA_K, iterations=50,
                                               reproj_err=1.0)
    assert ok == ok0
    if ok:
        assert np.array_equal(inliers.ravel(), inl0)


def test_invalid_arguments(posehip):
    import poseestimation_amd as pa
    obj, img, R, t, _ = datagen.pnp_scene(n=4, outlier_frac=0.0, seed=1)
    with pytest.raises(posehip.PosehipError):
        pa.solvePnPRansac(obj, img, datagen.CAMERA_K, None)          # n < 5
    obj, img, R, t, _ = datagen.pnp_scene(n=20, outlier_frac=0.0, seed=1)
    with pytest.raises(posehip.PosehipError):
        pa.solvePnPRansac(obj, img, datagen.CAMERA_K, None, confidence=1.0)
    with pytest.raises(ValueError):
        pa.solvePnPRansac(obj, img, datagen.CAMERA_K, np.array([0.1, 0, 0, 0]))


def test_batched_hypotheses_match_oracle(posehip):
    """Three problems of different sizes in one launch: every problem's models and
    counts equal the oracle's for that problem alone."""
    import torch
    dev = torch.device("cuda", 0)
    sizes, H = [700, 2000, 333], 96
    objs, imgs, subs = [], [], []
    for p, n in enumerate(sizes):
        obj, img, R, t, _ = datagen.pnp_scene(n=n, seed=20 + p, angle_deg=20.0 + 15 * p)
        objs.append(obj)
        imgs.append(img)
        subs.append(orc.ransac_subsets(n, H, rng_state=1234 + p))
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    d_obj = torch.from_numpy(np.concatenate(objs)).to(dev)
    d_img = torch.from_numpy(np.concatenate(imgs)).to(dev)
    d_sub = torch.from_numpy(np.concatenate(subs)).to(dev)
    d_off = torch.from_numpy(offsets).to(dev)
    d_models = torch.zeros((len(sizes) * H, 6), dtype=torch.float64, device=dev)
    d_counts = torch.zeros(len(sizes) * H, dtype=torch.int32, device=dev)
    Kc = np.ascontiguousarray(datagen.CAMERA_K, np.float64).reshape(9)
    torch.cuda.synchronize()
    posehip.check(posehip.load().ph_ransac_epnp_hypotheses_batched_dev(
        d_obj.data_ptr(), d_img.data_ptr(), d_off.data_ptr(), len(sizes),
        Kc.ctypes.data_as(C.POINTER(C.c_double)), d_sub.data_ptr(), H, 8.0, d_models.data_ptr(),
        d_counts.data_ptr(), None))
    posehip.check(posehip.load().ph_stream_synchronize(None))
    models, counts = d_models.cpu().numpy(), d_counts.cpu().numpy()
    for p in range(len(sizes)):
        rm, rc = orc.pnp_hypotheses(objs[p], imgs[p], datagen.CAMERA_K, subs[p])
        assert np.array_equal(counts[p * H:(p + 1) * H], rc)
        assert np.abs(models[p * H:(p + 1) * H] - rm).max() < 1e-12


def test_scene_to_wireframe_round_trip(posehip):
    """The whole glue around the pose stage: reference-frame keypoints lifted to 3D on the
    box (the TODO of src/cereal_box.cpp:292-295), matched pixels of a rotated frame,
    solvePnPRansac, and the GL model matrix for the wireframe (src/cereal_box.cpp:286)."""
    import poseestimation_amd as pa
    from poseestimation_amd import geometry as g
    rng = np.random.default_rng(7)
    K = g.camera_matrix()
    # two faces well in view: with a near-frontal view almost every keypoint lies on one
    # face and 5-point EPnP samples are coplanar (the degenerate case of the method)
    M_ref, M_cur = g.model_matrix(0.8), g.model_matrix(1.1)
    R_ref, t_ref = g.pose_from_model_matrix(M_ref)
    R_cur, t_cur = g.pose_from_model_matrix(M_cur)
    # keypoints of the reference frame: random pixels, keep those that hit the box
    px = rng.random((8000, 2)) * [800, 600]
    pts3d, hit = g.lift_pixels_to_box(px, R_ref, t_ref, K)
    pts3d = pts3d[hit][:600]
    assert len(pts3d) == 600
    # where those surface points appear in the current frame (+ noise, 20 % wrong matches)
    img = g.project_cv(pts3d, R_cur, t_cur, K) + rng.normal(0, 0.5, (600, 2))
    bad = rng.random(600) < 0.2
    img[bad] = rng.random((int(bad.sum()), 2)) * [800, 600]
    ok, rvec, tvec, inliers = pa.solvePnPRansac(pts3d, img, K, None, iterationsCount=200)
    assert ok and len(inliers) >= 0.75 * 600
    M_est = g.model_matrix_from_pose(rvec, tvec)
    assert np.abs(M_est - M_cur).max() < 1e-2
    ok0, r0, t0, inl0, _ = orc.solvepnp_ransac(pts3d, img, K, iterations=200)
    assert np.array_equal(inliers.ravel(), inl0)
    assert_same_pose(rvec, tvec, r0, t0)


def test_batched_solvepnp_ransac_equals_single_calls_and_oracle(posehip):
    """A ragged batch of frames in one call: every frame must get what its own
    solvePnPRansac call gets (bit for bit: same kernels, same order of sums) and what
    the oracle gets (inliers and iteration counts exactly, pose within 1e-4)."""
    import poseestimation_amd as pa
    specs = [(2000, 0.3, 3), (500, 0.5, 8), (64, 0.1, 9), (5, 0.0, 11), (300, 1.0, 12), (3, 0.0, 13),
             (800, 0.2, 14), (1000, 0.6, 15), (6, 0.0, 16)]
    scenes = [datagen.pnp_scene(n=n, outlier_frac=o, seed=s, noise_px=0.0 if n <= 6 else 1.0)
              for n, o, s in specs]
    objs, imgs = [sc[0] for sc in scenes], [sc[1] for sc in scenes]
    for iters in (100, 1000):
        res = pa.solvePnPRansacBatch(objs, imgs, datagen.CAMERA_K, None, iterationsCount=iters,
                                     returnIterations=True)
        assert len(res) == len(specs)
        for (n, o, s), obj, img, (ok, rvec, tvec, inl, it) in zip(specs, objs, imgs, res):
            if n < 5:
                assert not ok and inl is None and it == 0
                continue
            ok1, r1, t1, inl1 = pa.solvePnPRansac(obj, img, datagen.CAMERA_K, None,
                                                  iterationsCount=iters)
            assert ok == ok1
            if ok:
                assert np.array_equal(inl, inl1)
                assert np.array_equal(rvec, r1) and np.array_equal(tvec, t1)
            if n == 5:
                assert ok and inl.ravel().tolist() == [0, 1, 2, 3, 4] and it == 0
                continue
            ok0, r0, t0, inl0, it0 = orc.solvepnp_ransac(obj, img, datagen.CAMERA_K,
                                                         iterations=iters)
            assert ok == ok0 and it == it0
            if ok:
                assert np.array_equal(inl.ravel(), inl0)
                assert_same_pose(rvec, tvec, r0, t0)


def test_batched_solvepnp_ransac_many_small_frames(posehip):
    """64 frames of the reference's size (500 correspondences, 100 iterations): more
    live problems than one round's hypothesis budget per problem, several rounds."""
    import poseestimation_amd as pa
    scenes = [datagen.pnp_scene(n=500, outlier_frac=0.25 + 0.005 * f, seed=100 + f)
              for f in range(64)]
    res = pa.solvePnPRansacBatch([s[0] for s in scenes], [s[1] for s in scenes], datagen.CAMERA_K,
                                 None, iterationsCount=1000, returnIterations=True)
    for f in (0, 17, 63):
        obj, img = scenes[f][0], scenes[f][1]
        ok0, r0, t0, inl0, it0 = orc.solvepnp_ransac(obj, img, datagen.CAMERA_K, iterations=1000)
        ok, rvec, tvec, inl, it = res[f]
        assert ok == ok0 and it == it0 and np.array_equal(inl.ravel(), inl0)
        assert_same_pose(rvec, tvec, r0, t0)
    for (ok, rvec, tvec, inl, it), sc in zip(res, scenes):
        assert ok
        assert rel_rotation_error(orc.rodrigues(rvec.ravel()), sc[2]) < 2e-2
        assert rel_translation_error(tvec.ravel(), sc[3]) < 2e-2


def test_batched_solvepnp_ransac_invalid_arguments(posehip):
    lib = posehip.load()
    fp, dp, ip = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int32)
    obj, img = np.zeros((10, 3), np.float32), np.zeros((10, 2), np.float32)
    K = np.ascontiguousarray(datagen.CAMERA_K).reshape(9)
    out = np.zeros(6)
    i4 = np.zeros(10, np.int32)
    bad = np.array([0, 10, 5], np.int32)
    st = lib.ph_solvepnp_ransac_epnp_batched(obj.ctypes.data_as(fp), img.ctypes.data_as(fp),
                                             bad.ctypes.data_as(ip), 2, K.ctypes.data_as(dp), 100,
                                             8.0, 0.99, orc.CV_RNG_DEFAULT, out.ctypes.data_as(dp),
                                             out.ctypes.data_as(dp), i4.ctypes.data_as(ip),
                                             i4.ctypes.data_as(ip), i4.ctypes.data_as(ip), None)
    assert st == 1 and b"ascending" in lib.ph_last_error_string()
    ok = np.array([0, 10], np.int32)
    st = lib.ph_solvepnp_ransac_epnp_batched(obj.ctypes.data_as(fp), img.ctypes.data_as(fp),
                                             ok.ctypes.data_as(ip), 1, K.ctypes.data_as(dp), 100,
                                             8.0, 1.5, orc.CV_RNG_DEFAULT, out.ctypes.data_as(dp),
                                             out.ctypes.data_as(dp), i4.ctypes.data_as(ip),
                                             i4.ctypes.data_as(ip), i4.ctypes.data_as(ip), None)
    assert st == 1   # confidence outside (0, 1)


def test_hypotheses_beyond_one_gpu_full_equal_separate_calls(posehip):
    """More hypotheses than one round of solves (256 CUs x 128): the call splits into
    rounds and scores them on a second stream; every model and count must equal what
    two smaller calls (one round each, no second stream) produce, and a sample must
    equal the oracle."""
    H = 40000
    obj, img, R, t, _ = datagen.pnp_scene(n=300, outlier_frac=0.3, seed=31)
    sub = orc.ransac_subsets(300, H)
    for _ in range(2):   # twice: the second stream and its events are reused across calls
        m_all, c_all = gpu_hypotheses(posehip, obj, img, datagen.CAMERA_K, sub)
    m_a, c_a = gpu_hypotheses(posehip, obj, img, datagen.CAMERA_K, sub[:20000])
    m_b, c_b = gpu_hypotheses(posehip, obj, img, datagen.CAMERA_K, sub[20000:])
    assert np.array_equal(c_all, np.concatenate([c_a, c_b]))
    assert np.array_equal(m_all, np.concatenate([m_a, m_b]))
    pick = np.r_[0:64, 32760:32776, 39936:40000]
    rm, rc = orc.pnp_hypotheses(obj, img, datagen.CAMERA_K, sub[pick], threads=8)
    assert np.array_equal(c_all[pick], rc)
    assert np.abs(m_all[pick] - rm).max() < 1e-12


def test_degenerate_samples_agree_with_oracle(posehip):
    """Collinear, coincident and all-zero correspondences: the solve must terminate (its
    Jacobi sweeps and zero-row completion are bounded) and give the oracle's counts; models
    are compared where the oracle's are finite (NaN stays NaN)."""
    rng = np.random.default_rng(41)
    n = 64
    line = np.outer(np.linspace(-1, 1, n), [1.0, 0.5, 0.2]).astype(np.float32)
    same = np.tile(np.array([[0.3, -0.2, 0.1]], np.float32), (n, 1))
    zero = np.zeros((n, 3), np.float32)
    img = (rng.random((n, 2)) * [800, 600]).astype(np.float32)
    sub = orc.ransac_subsets(n, 32)
    for obj in (line, same, zero):
        m, c = gpu_hypotheses(posehip, obj, img, datagen.CAMERA_K, sub)
        rm, rc = orc.pnp_hypotheses(obj, img, datagen.CAMERA_K, sub)
        assert np.array_equal(c, rc)
        assert np.array_equal(np.isnan(m), np.isnan(rm))
        fin = np.isfinite(rm)
        assert np.allclose(m[fin], rm[fin], rtol=1e-9, atol=1e-9)


def test_every_solve_kernel_gives_the_same_models(posehip):
    """The library picks the solve kernel by the number of hypotheses in the call (three
    waves per 4 or per 8 hypotheses for a lone small call, one wave per 4, 8, 16 or 32
    beyond); every one performs the sequential statement's operations in its order, so a
    hypothesis must get bit for bit the same model and count whatever the call's size, the
    degenerate samples' NaNs included, and the oracle's counts on a sample."""
    obj, img, R, t, _ = datagen.pnp_scene(n=400, outlier_frac=0.3, seed=57)
    obj[:40] = obj[0]                      # some degenerate samples
    sub = orc.ransac_subsets(400, 17000)
    m_ref, c_ref = gpu_hypotheses(posehip, obj, img, datagen.CAMERA_K, sub)       # 32 per wave
    for h in (3, 100, 1021, 1024, 1025, 2048, 2049, 4096, 4097, 8192, 8193, 16384):
        m, c = gpu_hypotheses(posehip, obj, img, datagen.CAMERA_K, sub[:h])
        assert np.array_equal(c, c_ref[:h]), h
        assert np.array_equal(m.view(np.uint64), m_ref[:h].view(np.uint64)), h
    rm, rc = orc.pnp_hypotheses(obj, img, datagen.CAMERA_K, sub[:256], threads=8)
    assert np.array_equal(c_ref[:256], rc)
    assert np.array_equal(np.isnan(m_ref[:256]), np.isnan(rm))
    # the two-kernel forms of the large-batch solve (kept selectable; the single kernel is
    # the default) give the same bits, also across a round boundary (40 000 > 32 768)
    sub2 = orc.ransac_subsets(400, 40000)
    m1, c1 = gpu_hypotheses(posehip, obj, img, datagen.CAMERA_K, sub2)
    try:
        for mode in (2, 3):
            posehip.check(posehip.load().ph_ransac_set_solve_path(mode))
            m2, c2 = gpu_hypotheses(posehip, obj, img, datagen.CAMERA_K, sub2)
            assert np.array_equal(c2, c1), mode
            assert np.array_equal(m2.view(np.uint64), m1.view(np.uint64)), mode
    finally:
        posehip.load().ph_ransac_set_solve_path(0)


@pytest.mark.parametrize("n", [100_000, 1_000_000])
def test_solvepnp_ransac_many_correspondences(posehip, n):
    """Far beyond the reference's few hundred matches: the scoring kernel, the inlier list and
    the refit on hundreds of thousands of rows against the oracle (inliers and iteration
    count exactly, pose within 1e-4)."""
    import poseestimation_amd as pa
    obj, img, R, t, _ = datagen.pnp_scene(n=n, outlier_frac=0.3, seed=77)
    ok, rvec, tvec, inl = pa.solvePnPRansac(obj, img, datagen.CAMERA_K, None, iterationsCount=50)
    ok0, r0, t0, inl0, _ = orc.solvepnp_ransac(obj, img, datagen.CAMERA_K, iterations=50)
    assert ok and ok0 and len(inl0) > 0.6 * n
    assert np.array_equal(inl.ravel(), inl0)
    assert_same_pose(rvec, tvec, r0, t0)


def test_rows_on_the_threshold_circle_and_at_zero_depth(posehip):
    """The scoring kernel settles most (row, model) verdicts in packed fp32 with an error
    bound and falls back to the exact evaluation where the bound cannot decide.  Here the
    fallback is forced deterministically: rows that no sample uses are moved so that, for one
    model each, their reprojection lands on the threshold circle to within float rounding
    (and one ulp either side of it), or their depth in the camera frame is ~0, 0 or slightly
    negative.  Two problems with an ODD hypothesis count go down in one batched call, so that
    a wave's pair of hypotheses also straddles the two problems.  Every count must equal the
    oracle's."""
    import torch
    dev = torch.device("cuda", 0)
    K = datagen.CAMERA_K
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    H, thr = 65, 8.0
    objs, imgs, subs = [], [], []
    for p, n in enumerate([2000, 1501]):
        obj, img, _, _, _ = datagen.pnp_scene(n=n, seed=70 + p, angle_deg=25.0 + 20 * p)
        sub = orc.ransac_subsets(n, H, rng_state=77 + p)
        models, _ = orc.pnp_hypotheses(obj, img, K, sub)    # the rows below are in no sample,
        free = np.setdiff1d(np.arange(n), sub.ravel())       # so the models stay what they are
        usable = [h for h in range(H) if np.isfinite(models[h]).all()]
        rng = np.random.default_rng(90 + p)
        for j, row in enumerate(free):
            m = models[usable[j % len(usable)]]
            R, t = orc.rodrigues(m[:3]), m[3:]
            if j % 5 == 4:      # a point at (almost) no depth for this model
                eps = [1e-7, -1e-7, 1e-30, 0.0, 3e-5][(j // 5) % 5]
                cam = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), eps])
                obj[row] = (R.T @ (cam - t)).astype(np.float32)
                continue
            X = obj[row].astype(np.float64)
            c = R @ X + t
            u = np.array([fx * c[0] / c[2] + cx, fy * c[1] / c[2] + cy])
            th = rng.uniform(0, 2 * np.pi)
            px = (u + thr * np.array([np.cos(th), np.sin(th)])).astype(np.float32)
            step = [0, 1, -1, 2][j % 4]      # on the circle, and an ulp or two off it
            for _ in range(abs(step)):
                px[0] = np.nextafter(px[0], np.float32(np.inf if step > 0 else -np.inf))
            img[row] = px
        objs.append(obj)
        imgs.append(img)
        subs.append(sub)
    sizes = [o.shape[0] for o in objs]
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    d_obj = torch.from_numpy(np.concatenate(objs)).to(dev)
    d_img = torch.from_numpy(np.concatenate(imgs)).to(dev)
    d_sub = torch.from_numpy(np.concatenate(subs)).to(dev)
    d_off = torch.from_numpy(offsets).to(dev)
    d_models = torch.zeros((2 * H, 6), dtype=torch.float64, device=dev)
    d_counts = torch.zeros(2 * H, dtype=torch.int32, device=dev)
    Kc = np.ascontiguousarray(K, np.float64).reshape(9)
    torch.cuda.synchronize()
    posehip.check(posehip.load().ph_ransac_epnp_hypotheses_batched_dev(
        d_obj.data_ptr(), d_img.data_ptr(), d_off.data_ptr(), 2,
        Kc.ctypes.data_as(C.POINTER(C.c_double)), d_sub.data_ptr(), H, thr, d_models.data_ptr(),
        d_counts.data_ptr(), None))
    posehip.check(posehip.load().ph_stream_synchronize(None))
    models, counts = d_models.cpu().numpy(), d_counts.cpu().numpy()
    on_circle = 0
    for p in range(2):
        rm, rc = orc.pnp_hypotheses(objs[p], imgs[p], K, subs[p])
        assert np.array_equal(counts[p * H:(p + 1) * H], rc)
        assert_models_equal(models[p * H:(p + 1) * H], rm)
        # the construction did put rows on the circle: for a usable model, some row's exact
        # squared error is within 1e-4 of thr^2 (the fp32 pass's bound is far wider than that)
        h = next(h for h in range(H) if np.isfinite(rm[h]).all())
        R, t = orc.rodrigues(rm[h][:3]), rm[h][3:]
        c = objs[p].astype(np.float64) @ R.T + t
        with np.errstate(all="ignore"):
            u = np.stack([fx * c[:, 0] / c[:, 2] + cx, fy * c[:, 1] / c[:, 2] + cy], axis=1)
            err = ((u - imgs[p]) ** 2).sum(axis=1)
        on_circle += int((np.abs(err - thr * thr) < 1e-4 * thr * thr).sum())
    assert on_circle >= 10
