"""Host geometry that connects the reference's OpenGL scene to the PnP stage.

The reference renders a 2 x 2.5868 x 0.823 box (src/cereal_box.cpp:98-100) rotating
about (0.3, 1, 0) (:263) with view = translate(0, 0, -6) (:219) and
projection = perspective(45 deg, 800/600, 0.1, 100) (:222), reads the frame back
with glReadPixels (:279, row 0 = BOTTOM of the viewport) and leaves a TODO
(:292-295): "map from pixel location to 3D point in model coordinates (inverse of
projection*view*model)" so that reference-image keypoints get the 3D coordinates
EPnP needs, and a "TEMPORARY" ground-truth model matrix on the wireframe (:286)
that the estimated pose should replace.  This module is that glue, in plain numpy:

  camera_matrix()              pinhole K equivalent to the GL projection
  pose_from_model_matrix()     ground-truth (R, t) in OpenCV's camera frame
  model_matrix_from_pose()     estimated (rvec, tvec) -> GL model matrix for the wireframe
  lift_pixels_to_box()         pixel -> 3D point on the box surface (the TODO)

OpenCV's camera frame is x right, y down, z forward; GL eye space is x right,
y up, z backward: eye = diag(1, -1, -1) * camera.
"""
import numpy as np

SCR_WIDTH, SCR_HEIGHT = 800, 600                    # src/cereal_box.cpp:30-31
BOX_HALF_EXTENTS = np.array([1.0, 1.2934, 0.41149])  # src/cereal_box.cpp:98-100
_FLIP = np.diag([1.0, -1.0, -1.0])


def rodrigues(x):
    """Rotation vector (3,) -> matrix (3,3), or matrix -> vector (cv::Rodrigues, no Jacobian)."""
    x = np.asarray(x, dtype=np.float64)
    if x.size == 3:
        r = x.reshape(3)
        theta = np.linalg.norm(r)
        if theta < np.finfo(float).eps:
            return np.eye(3)
        k = r / theta
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        return np.cos(theta) * np.eye(3) + (1 - np.cos(theta)) * np.outer(k, k) + np.sin(theta) * Kx
    R = x.reshape(3, 3)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    axis = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.linalg.norm(axis) / 2
    c = np.clip((np.trace(R) - 1) / 2, -1.0, 1.0)
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        d = np.sqrt(np.maximum((np.diag(R) + 1) / 2, 0))
        d[1] *= -1.0 if R[0, 1] < 0 else 1.0
        d[2] *= -1.0 if R[0, 2] < 0 else 1.0
        if abs(d[0]) < abs(d[1]) and abs(d[0]) < abs(d[2]) and (R[1, 2] > 0) != (d[1] * d[2] > 0):
            d[2] = -d[2]
        return d * (theta / np.linalg.norm(d))
    return axis * (theta / (2 * s))


def perspective(fovy_deg=45.0, aspect=SCR_WIDTH / SCR_HEIGHT, near=0.1, far=100.0):
    """glm::perspective (right-handed, depth -1..1)."""
    f = 1.0 / np.tan(np.radians(fovy_deg) / 2)
    P = np.zeros((4, 4))
    P[0, 0] = f / aspect
    P[1, 1] = f
    P[2, 2] = (far + near) / (near - far)
    P[2, 3] = 2 * far * near / (near - far)
    P[3, 2] = -1.0
    return P


def view_matrix():
    """glm::translate(I, (0, 0, -6)): src/cereal_box.cpp:219."""
    V = np.eye(4)
    V[2, 3] = -6.0
    return V


def model_matrix(angle_rad, axis=(0.3, 1.0, 0.0)):
    """glm::rotate(I, angle, axis): src/cereal_box.cpp:263."""
    M = np.eye(4)
    a = np.asarray(axis, float)
    M[:3, :3] = rodrigues(a / np.linalg.norm(a) * angle_rad)
    return M


def camera_matrix(width=SCR_WIDTH, height=SCR_HEIGHT, fovy_deg=45.0, pixel_centers=False):
    """Pinhole K of the GL projection, for pixel rows counted from the TOP (OpenCV's
    convention): fx = fy = (height/2) / tan(fovy/2), principal point at the centre of
    the window in continuous window coordinates (400, 300).  With pixel_centers=True
    the principal point is given in OpenCV keypoint coordinates, where the centre of
    pixel (i, j) is (i, j) and not (i + 0.5, j + 0.5): (399.5, 299.5)."""
    f = (height / 2.0) / np.tan(np.radians(fovy_deg) / 2)
    off = 0.5 if pixel_centers else 0.0
    return np.array([[f, 0.0, width / 2.0 - off], [0.0, f, height / 2.0 - off], [0.0, 0.0, 1.0]])


def readback_to_top_down(pixels, height=SCR_HEIGHT):
    """glReadPixels rows start at the bottom of the viewport (the reference runs ORB on
    that unflipped frame); convert (u, v_from_bottom) to (u, v_from_top) pixel centres."""
    p = np.array(pixels, dtype=np.float64).reshape(-1, 2)
    p[:, 1] = (height - 1) - p[:, 1]
    return p


def pose_from_model_matrix(model, view=None):
    """(R, t): model coordinates -> OpenCV camera frame, from the GL matrices."""
    view = view_matrix() if view is None else view
    mv = view @ model
    return _FLIP @ mv[:3, :3], _FLIP @ mv[:3, 3]


def model_matrix_from_pose(rvec, tvec, view=None):
    """GL model matrix that draws the box at the estimated pose (replaces the ground
    truth handed to the wire shader at src/cereal_box.cpp:286)."""
    view = view_matrix() if view is None else view
    mv = np.eye(4)
    mv[:3, :3] = _FLIP @ rodrigues(np.ravel(rvec))
    mv[:3, 3] = _FLIP @ np.ravel(tvec)
    return np.linalg.inv(view) @ mv


def project_gl(points, model, view=None, proj=None, width=SCR_WIDTH, height=SCR_HEIGHT):
    """Window coordinates (x right, y UP, pixel centres at +0.5) the GL pipeline gives."""
    view = view_matrix() if view is None else view
    proj = perspective() if proj is None else proj
    X = np.c_[np.asarray(points, float), np.ones(len(points))]
    clip = X @ (proj @ view @ model).T
    ndc = clip[:, :3] / clip[:, 3:4]
    return np.c_[(ndc[:, 0] + 1) * width / 2, (ndc[:, 1] + 1) * height / 2]


def project_cv(points, R, t, K):
    """Pinhole projection, pixel rows from the top."""
    cam = np.asarray(points, float) @ np.asarray(R).T + np.asarray(t)
    return cam[:, :2] / cam[:, 2:3] * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]])


def lift_pixels_to_box(pixels, R, t, K, half_extents=BOX_HALF_EXTENTS):
    """The TODO of src/cereal_box.cpp:292-295: 3D model coordinates of the box surface
    seen at each pixel (rows from the top), given the pose (R, t) of the reference
    frame.  Returns (points[n, 3], hit[n]); rays that miss the box get hit = False.
    """
    px = np.asarray(pixels, dtype=np.float64).reshape(-1, 2)
    R = np.asarray(R, float)
    t = np.asarray(t, float).reshape(3)
    rays_cam = np.c_[(px[:, 0] - K[0, 2]) / K[0, 0], (px[:, 1] - K[1, 2]) / K[1, 1], np.ones(len(px))]
    origin = -R.T @ t                 # camera centre in model coordinates
    dirs = rays_cam @ R               # R^T applied to every ray
    h = np.asarray(half_extents, float)
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (-h - origin) / dirs
        t2 = (h - origin) / dirs
    lo = np.fmax.reduce(np.minimum(t1, t2), axis=1)     # slab method; fmax/fmin skip NaN
    hi = np.fmin.reduce(np.maximum(t1, t2), axis=1)
    hit = (lo <= hi) & (hi > 0)
    s = np.where(lo > 0, lo, hi)
    pts = origin + dirs * s[:, None]
    pts[~hit] = np.nan
    return pts, hit
