"""Generates tests/golden/evaluation.json by importing the reference's own
src/evaluation/Evaluation.py (run in the build container only; the reference does
not travel to the GPU box).  Inputs: the demo values of Evaluation.py:48-51,62-65,
76-78 plus seeded random poses."""
import importlib.util
import json
import os
import sys

import numpy as np

REF = "/root/reference/src/evaluation/Evaluation.py"
spec = importlib.util.spec_from_file_location("ref_evaluation", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def rot(axis, angle):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    x, y, z = axis
    K = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
    return np.cos(angle) * np.eye(3) + (1 - np.cos(angle)) * np.outer(axis, axis) + np.sin(angle) * K


cases = []
# the reference's own demo
t_gt0 = np.array([3, 2, 1]); t_est1 = np.array([[3, 2, 1]]); t_est2 = np.array([[4, 5, 6]])
r_gt0 = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]])
r_est1 = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]])
r_est2 = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]])
pts3d0 = np.array([[100, 100, 100], [5, 10, 10], [1, 2, 3], [2, 2, 2]])
for (r_est, t_est) in ((r_est1, t_est1), (r_est2, t_est2)):
    cases.append(dict(pts3d=pts3d0.tolist(), diameter=100, R_pred=r_est.tolist(), t_pred=t_est.tolist(),
                      R_gt=r_gt0.tolist(), t_gt=t_gt0.tolist(),
                      translation_error=float(ref.translation_error(t_est, t_gt0)),
                      rotation_error=float(ref.rotation_error(r_est, r_gt0)),
                      add=float(ref.compute_add_score(pts3d0, 100, r_est, t_est, r_gt0, t_gt0)),
                      adds=float(ref.compute_adds_score(pts3d0, 100, r_est, t_est, r_gt0, t_gt0))))
rng = np.random.default_rng(0)
for i in range(6):
    R_gt = rot(rng.normal(size=3), rng.uniform(0, 3))
    R_pred = rot(rng.normal(size=3), rng.uniform(0, 0.2)) @ R_gt
    t_gt = rng.normal(size=3) * 3
    t_pred = (t_gt + rng.normal(size=3) * 0.3).reshape(1, 3)
    pts = rng.uniform(-2, 2, size=(12, 3))
    diameter = float(rng.uniform(2, 30))
    cases.append(dict(pts3d=pts.tolist(), diameter=diameter, R_pred=R_pred.tolist(), t_pred=t_pred.tolist(),
                      R_gt=R_gt.tolist(), t_gt=t_gt.tolist(),
                      translation_error=float(ref.translation_error(t_pred, t_gt)),
                      rotation_error=float(ref.rotation_error(R_pred, R_gt)),
                      add=float(ref.compute_add_score(pts, diameter, R_pred, t_pred, R_gt, t_gt)),
                      adds=float(ref.compute_adds_score(pts, diameter, R_pred, t_pred, R_gt, t_gt))))
# slightly and grossly non-orthogonal inputs: scipy's Rotation.from_matrix (1.15: nearest
# rotation by SVD) projects them before the reference compares; rotation error only
rng2 = np.random.default_rng(5)
nonorth = []
for eps in (1e-6, 1e-3, 1e-1):
    R_gt = rot(rng2.normal(size=3), 1.0)
    R_pred = rot(rng2.normal(size=3), 0.1) @ R_gt + rng2.normal(size=(3, 3)) * eps
    nonorth.append(dict(R_pred=R_pred.tolist(), R_gt=R_gt.tolist(),
                        rotation_error=float(ref.rotation_error(R_pred, R_gt))))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "evaluation.json")
json.dump({"_about": "outputs of the reference's src/evaluation/Evaluation.py (numpy %s)" % np.__version__,
           "cases": cases, "non_orthogonal_rotation_cases": nonorth}, open(out, "w"), indent=1)
print("wrote", out, len(cases), "cases")
