"""Seeded synthetic descriptor sets shared by the tests and the benchmark."""
import numpy as np


def random_descriptors(n, seed):
    """iid uniform bytes, uint8 [n, 32]."""
    return np.random.default_rng(seed).integers(0, 256, size=(n, 32), dtype=np.uint8)


def perturbed_copy(desc, seed, max_flips=40, shuffle=True):
    """A permuted copy of `desc` with 0..max_flips random bit flips per row, so
    that true correspondences exist (cross-check and the min-distance filter
    have something to keep)."""
    rng = np.random.default_rng(seed)
    out = desc.copy()
    n = out.shape[0]
    bits = np.unpackbits(out, axis=1)
    for i in range(n):
        k = int(rng.integers(0, max_flips + 1))
        pos = rng.choice(256, size=k, replace=False)
        bits[i, pos] ^= 1
    out = np.packbits(bits, axis=1)
    if shuffle:
        out = out[rng.permutation(n)]
    return np.ascontiguousarray(out)


def codebook_descriptors(n, seed, words=16):
    """Tie-stress set: every row drawn from a tiny codebook, so most distances
    are equal and the lowest-index rule decides nearly every match."""
    rng = np.random.default_rng(seed)
    book = rng.integers(0, 256, size=(words, 32), dtype=np.uint8)
    return np.ascontiguousarray(book[rng.integers(0, words, size=n)])


def numpy_knn(q, t, k):
    """Independent brute force in numpy: stable argsort gives lowest index on ties."""
    qb = np.unpackbits(q, axis=1).astype(np.int32)
    tb = np.unpackbits(t, axis=1).astype(np.int32)
    d = qb @ (1 - tb).T + (1 - qb) @ tb.T
    order = np.argsort(d, axis=1, kind="stable")[:, :k]
    idx = np.full((q.shape[0], k), -1, np.int32)
    dist = np.full((q.shape[0], k), 2147483647, np.int32)
    kk = order.shape[1]
    idx[:, :kk] = order
    dist[:, :kk] = np.take_along_axis(d, order, axis=1)
    return idx, dist


# ---- 2D-3D correspondences of the cereal-box scene --------------------------------
# Box half extents (src/cereal_box.cpp:98-100), camera of glm::perspective(45 deg,
# 800/600) (src/cereal_box.cpp:219-222): fy = 300 / tan(22.5 deg), square pixels.
BOX_HALF = np.array([1.0, 1.2934, 0.41149])
FOCAL = 300.0 / np.tan(np.radians(22.5))
CAMERA_K = np.array([[FOCAL, 0.0, 400.0], [0.0, FOCAL, 300.0], [0.0, 0.0, 1.0]])


def axis_angle_matrix(axis, angle):
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    x, y, z = axis
    c, s = np.cos(angle), np.sin(angle)
    Kx = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
    return c * np.eye(3) + (1 - c) * np.outer(axis, axis) + s * Kx


def pnp_scene(n=2000, outlier_frac=0.3, noise_px=1.0, seed=3, angle_deg=50.0 * 0.7,
              t=(0.0, 0.0, 6.0), on_faces=False):
    """(obj[n,3] f32, img[n,2] f32, R_gt, t_gt, is_inlier[n]).

    Points uniform in the box volume (or on its faces), pose = rotation about
    (0.3, 1, 0) (src/cereal_box.cpp:263) and the camera 6 units away; inliers get
    N(0, noise_px) pixel noise, outliers are uniform in the 800x600 frame.
    """
    rng = np.random.default_rng(seed)
    obj = (rng.random((n, 3)) * 2 - 1) * BOX_HALF
    if on_faces:
        face = rng.integers(0, 3, size=n)
        sign = rng.integers(0, 2, size=n) * 2 - 1
        obj[np.arange(n), face] = sign * BOX_HALF[face]
    R = axis_angle_matrix((0.3, 1.0, 0.0), np.radians(angle_deg))
    tv = np.asarray(t, np.float64)
    cam = obj @ R.T + tv
    px = cam[:, :2] / cam[:, 2:3] * FOCAL + np.array([400.0, 300.0])
    px += rng.normal(0.0, noise_px, size=px.shape)
    is_inlier = rng.random(n) >= outlier_frac
    out_px = rng.random((n, 2)) * np.array([800.0, 600.0])
    img = np.where(is_inlier[:, None], px, out_px)
    return obj.astype(np.float32), img.astype(np.float32), R, tv, is_inlier
