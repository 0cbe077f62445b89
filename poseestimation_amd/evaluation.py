"""Pose-error metrics of the reference's evaluation harness, restated.

Reference: src/evaluation/Evaluation.py:5-43 (translation_error, rotation_error,
compute_add_score, compute_adds_score).  Host-side numpy, like the reference; the
relative translation / rotation errors are the tolerance metric of the pose
parity tests ("1e-4 relative rotation/translation error").

compute_add_score / compute_adds_score reproduce the reference's arithmetic
exactly, including its use of `*` between a 3x3 rotation and a point, which
numpy evaluates as an element-wise broadcast (column c of R scaled by p[c]), not
as a matrix-vector product, and the transposed-translation broadcast that goes
with it.  They are NOT the textbook ADD / ADD-S; a caller who wants those should
not use these two functions.
"""
import numpy as np


def translation_error(t_est, t_gt):
    """||t_gt - t_est|| / ||t_est||  (Evaluation.py:5-7)."""
    t_est = np.asarray(t_est)
    t_gt = np.asarray(t_gt)
    return np.linalg.norm(t_gt - t_est) / np.linalg.norm(t_est)


def rotation_error(r_est, r_gt):
    """Frobenius ||R_gt - R_est|| / ||R_est||  (Evaluation.py:9-15).

    The reference passes both matrices through scipy's Rotation.from_matrix /
    as_matrix first.  For an input that is not orthogonal scipy (1.15) takes the nearest
    rotation by SVD, and so does this function (the golden file holds such cases, equal
    to 1e-12); for a determinant <= 0 scipy raises ValueError, and so does this function.
    """
    q_est = _nearest_rotation(r_est)
    q_gt = _nearest_rotation(r_gt)
    return np.linalg.norm(q_gt - q_est) / np.linalg.norm(q_est)


def _nearest_rotation(R):
    R = np.asarray(R, dtype=np.float64)
    if not np.linalg.det(R) > 0:
        raise ValueError("Non-positive determinant (left-handed or null coordinate frame) in "
                         "rotation matrix")
    if abs(np.linalg.det(R) - 1.0) < 1e-12 and np.allclose(R @ R.T, np.eye(3), atol=1e-12):
        return R
    U, _, Vt = np.linalg.svd(R)
    if np.linalg.det(U @ Vt) < 0:
        U[:, -1] = -U[:, -1]
    return U @ Vt


def _broadcast_pose(R, p, t):
    """R * p^T + t^T as numpy evaluates it for a 1-D point (Evaluation.py:22-23)."""
    return np.asarray(R) * np.asarray(p).transpose() + np.asarray(t).transpose()


def _score(mean_distances, diameter, percentage):
    threshold = diameter * percentage
    return (mean_distances < threshold).sum() / mean_distances.shape[0]


def compute_add_score(pts3d, diameter, R_pred, t_pred, R_gt, t_gt, percentage=0.1):
    """Evaluation.py:18-29."""
    pts3d = np.asarray(pts3d)
    mean_distances = np.zeros((pts3d.shape[0],), dtype=np.float32)
    for i, p in enumerate(pts3d):
        diff = _broadcast_pose(R_gt, p, t_gt) - _broadcast_pose(R_pred, p, t_pred)
        mean_distances[i] = np.mean(np.linalg.norm(diff, axis=0))
    return _score(mean_distances, diameter, percentage)


def compute_adds_score(pts3d, diameter, R_pred, t_pred, R_gt, t_gt, percentage=0.1):
    """Evaluation.py:32-43 (nearest neighbour by brute force instead of a KD-tree)."""
    pts3d = np.asarray(pts3d)
    mean_distances = np.zeros((pts3d.shape[0],), dtype=np.float32)
    for i, p in enumerate(pts3d):
        gt = _broadcast_pose(R_gt, p, t_gt).transpose()        # rows = reference "points"
        pred = _broadcast_pose(R_pred, p, t_pred).transpose()
        d = np.linalg.norm(pred[:, None, :] - gt[None, :, :], axis=2)
        mean_distances[i] = np.mean(d.min(axis=1))
    return _score(mean_distances, diameter, percentage)
