"""CPU suite: the GL <-> pinhole glue (the reference has only a TODO here, so the checks
are round trips and agreement between the GL pipeline and the pinhole model)."""
import numpy as np

from poseestimation_amd import geometry as g


def box_corners():
    h = g.BOX_HALF_EXTENTS
    return np.array([[sx * h[0], sy * h[1], sz * h[2]] for sx in (-1, 1) for sy in (-1, 1)
                     for sz in (-1, 1)])


def test_pinhole_equals_gl_pipeline():
    K = g.camera_matrix()
    assert abs(K[0, 0] - 724.2641) < 1e-3 and K[0, 2] == 400 and K[1, 2] == 300
    for angle in (0.0, 0.4, 1.3, 2.9):
        M = g.model_matrix(angle)
        R, t = g.pose_from_model_matrix(M)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and np.linalg.det(R) > 0
        win = g.project_gl(box_corners(), M)                # y up, continuous window coords
        cv = g.project_cv(box_corners(), R, t, K)           # y down
        assert np.allclose(win[:, 0], cv[:, 0], atol=1e-9)
        assert np.allclose(g.SCR_HEIGHT - win[:, 1], cv[:, 1], atol=1e-9)


def test_model_matrix_round_trip():
    M = g.model_matrix(0.77)
    R, t = g.pose_from_model_matrix(M)
    M2 = g.model_matrix_from_pose(g.rodrigues(R), t)
    assert np.allclose(M, M2, atol=1e-12)


def test_rodrigues_round_trip():
    rng = np.random.default_rng(0)
    for _ in range(20):
        r = rng.normal(size=3)
        r *= rng.uniform(0, np.pi - 1e-3) / np.linalg.norm(r)
        assert np.allclose(g.rodrigues(g.rodrigues(r)), r, atol=1e-9)
    assert np.allclose(g.rodrigues(np.zeros(3)), np.eye(3))


def test_lift_recovers_surface_points():
    rng = np.random.default_rng(1)
    K = g.camera_matrix()
    M = g.model_matrix(0.6)
    R, t = g.pose_from_model_matrix(M)
    h = g.BOX_HALF_EXTENTS
    # random points on the faces; keep the ones facing the camera
    pts = (rng.random((400, 3)) * 2 - 1) * h
    face = rng.integers(0, 3, 400)
    sign = rng.integers(0, 2, 400) * 2 - 1
    pts[np.arange(400), face] = sign * h[face]
    normals = np.zeros((400, 3))
    normals[np.arange(400), face] = sign
    cam_centre = -R.T @ t
    visible = np.einsum("ij,ij->i", normals, cam_centre - pts) > 1e-6
    px = g.project_cv(pts[visible], R, t, K)
    lifted, hit = g.lift_pixels_to_box(px, R, t, K)
    assert hit.all()
    assert np.allclose(lifted, pts[visible], atol=1e-9)
    # a pixel in the corner of the frame misses the box
    _, hit = g.lift_pixels_to_box([[2.0, 2.0]], R, t, K)
    assert not hit[0]


def test_readback_rows_are_bottom_up():
    p = g.readback_to_top_down([[10, 0], [10, 599]])
    assert p.tolist() == [[10.0, 599.0], [10.0, 0.0]]


def test_cpp_geometry_header_agrees_with_python(tmp_path):
    """include/posehip_geometry.hpp (the C++ side of the same glue, for the reference's
    C++ program) against this module: K, pose from the GL matrices, model matrix from a
    pose, and the pixel -> box-surface lifting, on the same inputs."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "geo.cpp"
    src.write_text(r'''
#include <cstdio>
#include "posehip_geometry.hpp"
using namespace posehip::geometry;
static void arr(const char *name, const double *a, int n, bool last = false)
{
    std::printf("\"%s\": [", name);
    for (int i = 0; i < n; i++)
        std::printf("%s%.17g", i ? ", " : "", std::isnan(a[i]) ? -12345.0 : a[i]);
    std::printf("]%s\n", last ? "" : ",");
}
int main()
{
    double K[9], Kc[9], P[16], V[16], M[16], R[9], t[3], M2[16];
    cameraMatrix(K);
    cameraMatrix(Kc, 800, 600, 45.0, true);
    perspective(45.0, 800.0 / 600.0, 0.1, 100.0, P);
    viewMatrix(V);
    const double axis[3] = {0.3, 1.0, 0.0};
    modelMatrix(0.9, axis, M);
    poseFromModelMatrix(M, V, R, t);
    const double rvec[3] = {0.2, -0.7, 0.4}, tvec[3] = {0.3, -0.2, 5.5};
    modelMatrixFromPose(rvec, tvec, V, M2);
    const double px[] = {400, 300, 10, 10, 420.5, 250.25, 330, 401, 799, 0, 380, 120, 455.5, 333.25};
    double pts[21];
    std::vector<unsigned char> hit = liftPixelsToBox(px, 7, R, t, K, pts);
    double h[7];
    for (int i = 0; i < 7; i++) h[i] = hit[i];
    std::printf("{");
    arr("K", K, 9); arr("Kc", Kc, 9); arr("P", P, 16); arr("V", V, 16); arr("M", M, 16);
    arr("R", R, 9); arr("t", t, 3); arr("M2", M2, 16); arr("pts", pts, 21); arr("hit", h, 7, true);
    std::printf("}\n");
    return 0;
}
''')
    exe = tmp_path / "geo"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror",
                           "-I" + os.path.join(root, "include"), str(src), "-o", str(exe)])
    out = json.loads(subprocess.check_output([str(exe)], text=True))
    cm = lambda a: np.array(a).reshape(4, 4).T          # column-major 16 -> matrix
    assert np.allclose(np.array(out["K"]).reshape(3, 3), g.camera_matrix(), atol=1e-12)
    assert np.allclose(np.array(out["Kc"]).reshape(3, 3), g.camera_matrix(pixel_centers=True), atol=1e-12)
    assert np.allclose(cm(out["P"]), g.perspective(), atol=1e-12)
    assert np.allclose(cm(out["V"]), g.view_matrix(), atol=0)
    M = g.model_matrix(0.9)
    assert np.allclose(cm(out["M"]), M, atol=1e-12)
    R, t = g.pose_from_model_matrix(M)
    assert np.allclose(np.array(out["R"]).reshape(3, 3), R, atol=1e-12)
    assert np.allclose(out["t"], t, atol=1e-12)
    M2 = g.model_matrix_from_pose([0.2, -0.7, 0.4], [0.3, -0.2, 5.5])
    assert np.allclose(cm(out["M2"]), M2, atol=1e-12)
    px = np.array([400, 300, 10, 10, 420.5, 250.25, 330, 401, 799, 0, 380, 120, 455.5, 333.25]).reshape(-1, 2)
    pts, hit = g.lift_pixels_to_box(px, R, t, g.camera_matrix())
    assert np.array_equal(np.array(out["hit"]).astype(bool), hit) and hit.any() and not hit.all()
    got = np.array(out["pts"]).reshape(-1, 3)
    assert np.allclose(got[hit], pts[hit], atol=1e-10)
    assert (got[~hit] == -12345.0).all()
