"""Metrics of the reference (`src/shoulder/humerus/bone_props.py`), oracle side (test infrastructure).
  side              :12-47   groove side w.r.t. the head axis in construct_csys(canal, head-central)
  retroversion      :50-85   angle of the neck-normal axis in construct_csys(canal, TE); NOTE the reference
                             feeds `axis_normal()` in the CURRENT coordinate system (:72-73), not in CT
  neckshaft         :88-112
  radius_curvature  :115-148 least-squares sphere through the articular mask points (OBB frame)
  unitxyz_to_spherical  utils.py:321-332
`spherefit` / `unitxyz_to_spherical` are pinned by tests/golden/metrics_golden.npz; side / retroversion / neckshaft by
tests/golden/metrics_landmarks_golden.npz (the reference's own Side / RetroVersion / NeckShaft classes run on stand-in
landmark objects, tests/golden/make_metrics_golden.py)."""
import numpy as np

from .xform import construct_csys, transform_pts, unit_vector


def unitxyz_to_spherical(xyz):
    r = np.sqrt(np.sum(xyz ** 2))
    theta = np.rad2deg(np.arctan2(xyz[1], xyz[0]))
    phi = np.rad2deg(np.arccos(xyz[2] / r))
    return np.array([r, theta, phi])


def side(canal_axis_ct, central_axis_ct, groove_points_ct):
    T = construct_csys(canal_axis_ct, central_axis_ct)
    bg = np.mean(transform_pts(groove_points_ct, T), axis=0)
    return "left" if bg[1] <= 0 else "right"


def retroversion(canal_axis_ct, te_axis_ct, axis_normal_current, side_str):
    T = construct_csys(canal_axis_ct, te_axis_ct)
    an = transform_pts(axis_normal_current, T)
    an = unit_vector(an[0], an[1])
    an[0] = -1 * an[0]
    theta = unitxyz_to_spherical(an)[1]
    if side_str == "right":
        theta *= -1
    return float(theta)


def neckshaft(canal_axis_ct, normal_axis_ct):
    T = construct_csys(canal_axis_ct, normal_axis_ct)
    an = transform_pts(normal_axis_ct, T)
    an = unit_vector(an[0], an[1])
    return float(180 - unitxyz_to_spherical(an)[2])


def spherefit(pts):
    A = np.zeros((len(pts), 4))
    A[:, :3] = pts * 2
    A[:, 3] = 1
    f = np.zeros((len(pts), 1))
    f[:, 0] = (pts * pts).sum(axis=1)
    C, _, _, _ = np.linalg.lstsq(A, f, rcond=None)
    t = (C[0] * C[0]) + (C[1] * C[1]) + (C[2] * C[2]) + C[3]
    return float(np.sqrt(t)[0]), C[:-1].reshape(-1, 3)
