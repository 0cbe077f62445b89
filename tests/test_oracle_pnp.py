"""CPU suite: the oracle's RANSAC-EPnP restatement, pinned by properties
(the reference ships no fixture for this path: parity unpinned)."""
import numpy as np
import pytest

import datagen
import oracle_bind as orc


def cv_rng_python(state, count):
    """cv::RNG written out independently: state = lo32 * 4164903690 + hi32."""
    out = []
    for _ in range(count):
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & 0xFFFFFFFFFFFFFFFF
        out.append(state & 0xFFFFFFFF)
    return out


def test_subsets_follow_cv_rng_and_redraw_duplicates():
    n_points = 7  # few points: duplicates are common, so the redraw rule is exercised
    draws = iter(cv_rng_python(0xFFFFFFFFFFFFFFFF, 10000))
    want = []
    for _ in range(200):
        idx = []
        for i in range(5):
            v = next(draws) % n_points
            while v in idx:
                v = next(draws) % n_points
            idx.append(v)
        want.append(idx)
    got = orc.ransac_subsets(n_points, 200)
    assert got.tolist() == want


def test_update_num_iters():
    # log(1-0.99)/log(1-(0.7)^5) = 25.03 -> 25
    assert orc.update_num_iters(0.99, 0.3, 5, 4096) == 25
    assert orc.update_num_iters(0.99, 0.3, 5, 10) == 10      # never increases
    assert orc.update_num_iters(0.99, 0.0, 5, 100) == 0      # all inliers: denom = log(0)
    assert orc.update_num_iters(0.99, 1.0, 5, 100) == 100    # no inliers
    assert orc.update_num_iters(0.99, 0.9975, 5, 4096) == 4096


@pytest.mark.parametrize("angle", [0.0, 1e-9, 0.3, 1.0, 3.0, np.pi - 1e-9, np.pi])
def test_rodrigues_round_trip(angle):
    R = datagen.axis_angle_matrix((0.3, 1.0, 0.0), angle)
    r = orc.rodrigues(R)
    assert abs(np.linalg.norm(r) - angle) < 1e-7
    assert np.abs(orc.rodrigues(r) - R).max() < 1e-7


@pytest.mark.parametrize("n", [5, 6, 10, 100, 1500])
def test_epnp_recovers_exact_pose(n):
    obj, img, R, t, _ = datagen.pnp_scene(n=n, outlier_frac=0.0, noise_px=0.0, seed=n)
    r, tv = orc.solvepnp_epnp(obj, img, datagen.CAMERA_K)
    # float32 inputs bound the accuracy
    assert np.abs(orc.rodrigues(r) - R).max() < 2e-5
    assert np.abs(tv - t).max() < 2e-4


def test_epnp_planar_points_do_not_crash():
    obj, img, R, t, _ = datagen.pnp_scene(n=40, outlier_frac=0.0, noise_px=0.0, seed=5)
    obj[:, 2] = 0.41149  # one face of the box: exactly planar
    cam = obj.astype(np.float64) @ R.T + t
    img = (cam[:, :2] / cam[:, 2:3] * datagen.FOCAL + [400, 300]).astype(np.float32)
    r, tv = orc.solvepnp_epnp(obj, img, datagen.CAMERA_K)
    assert np.all(np.isfinite(r)) and np.all(np.isfinite(tv))


def test_inlier_rule_is_squared_float_error():
    obj, img, R, t, _ = datagen.pnp_scene(n=200, outlier_frac=0.0, noise_px=0.0, seed=6)
    r = orc.rodrigues(R)
    img2 = img.copy()
    img2[0, 0] += 8.0          # exactly 8 px off: err = 64 <= 64 -> inlier
    img2[1, 0] += 8.01         # just outside
    img2[2] += np.float32(6.0)  # 6*sqrt(2) = 8.49 px -> outlier
    good, mask, err = orc.count_inliers(obj, img2, datagen.CAMERA_K, r, t, 8.0)
    assert mask[1] == 0 and mask[2] == 0 and mask[3:].all()
    assert good == int(mask.sum())
    assert err.dtype == np.float32


def test_ransac_recovers_pose_and_stops_early():
    obj, img, R, t, inl = datagen.pnp_scene(n=2000, outlier_frac=0.3, seed=3)
    ok, r, tv, inliers, iters = orc.solvepnp_ransac(obj, img, datagen.CAMERA_K, iterations=4096)
    assert ok and iters < 100            # adaptive budget: ~25 iterations at 70 % inliers
    assert abs(len(inliers) - inl.sum()) <= 0.02 * inl.sum()
    assert np.linalg.norm(orc.rodrigues(r) - R) / np.linalg.norm(R) < 1e-2
    assert np.linalg.norm(tv - t) / np.linalg.norm(t) < 1e-2
    assert np.all(np.diff(inliers) > 0)


def test_ransac_replay_equals_batch_hypotheses():
    """The sequential loop equals: all hypotheses scored independently + the
    'strictly larger count wins, budget shrinks' rule replayed over the counts."""
    obj, img, R, t, inl = datagen.pnp_scene(n=800, outlier_frac=0.4, seed=14)
    iters = 300
    subsets = orc.ransac_subsets(800, iters)
    models, counts = orc.pnp_hypotheses(obj, img, datagen.CAMERA_K, subsets)
    niters, best, max_good, it = iters, None, 0, 0
    while it < niters:
        if counts[it] > max(max_good, 4):
            max_good, best = int(counts[it]), it
            niters = orc.update_num_iters(0.99, (800 - max_good) / 800, 5, niters)
        it += 1
    ok, r, tv, inliers, iters_run = orc.solvepnp_ransac(obj, img, datagen.CAMERA_K,
                                                        iterations=iters)
    assert ok and iters_run == it and len(inliers) == max_good
    good, mask, _ = orc.count_inliers(obj, img, datagen.CAMERA_K, models[best, :3],
                                      models[best, 3:], 8.0)
    assert np.array_equal(np.nonzero(mask)[0], inliers)


def test_multithreaded_hypotheses_equal_single():
    obj, img, R, t, inl = datagen.pnp_scene(n=500, seed=15)
    subsets = orc.ransac_subsets(500, 64)
    a = orc.pnp_hypotheses(obj, img, datagen.CAMERA_K, subsets)
    b = orc.pnp_hypotheses(obj, img, datagen.CAMERA_K, subsets, threads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_ransac_final_solve_takes_the_consensus_set_as_f64_points():
    """solvePnPRansac converts its consensus set to Point3d / Point2d before the final
    solvePnP, so undistortPoints keeps the normalised image points in double there, while
    the caller's CV_32F points (and RANSAC's samples) are rounded to float on the way.
    The final pose equals the f64-points solve on the inliers bit for bit, and differs from
    the f32-points solve of the same rows by float rounding only."""
    obj, img, R, t, _ = datagen.pnp_scene(n=400, outlier_frac=0.3, seed=12)
    ok, rvec, tvec, inl, _ = orc.solvepnp_ransac(obj, img, datagen.CAMERA_K, iterations=100)
    assert ok and len(inl) >= 200
    r64, t64 = orc.solvepnp_epnp(obj[inl], img[inl], datagen.CAMERA_K, points_as_f64=True)
    assert np.array_equal(rvec, r64) and np.array_equal(tvec, t64)
    r32, t32 = orc.solvepnp_epnp(obj[inl], img[inl], datagen.CAMERA_K)
    assert not (np.array_equal(r32, r64) and np.array_equal(t32, t64))
    assert np.abs(r32 - r64).max() < 1e-6 and np.abs(t32 - t64).max() < 1e-5
