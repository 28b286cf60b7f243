"""K12-K17 + K11: bicipital groove (oracle; test infrastructure).

Restates reference `src/shoulder/humerus/bicipital_groove.py`:
  x_process        :29-158  (per-row savgol/find_peaks/features, StandardScaler)
  groove_points    :160-242 (RFC -> linear KDE argmax -> per-row local radius minimum)
  groove_axis      :244-265 (line fit; canonical B-4: direction_z > 0 so row 0 is proximal)
scipy.signal / sklearn are the same third-party code the reference calls.  Pinned by
tests/golden/groove_golden.npz (reference's own DeepGroove.points() on injected
contours).  The quirks of SURVEY App. H are kept: canal axis in CT coordinates mixed
with OBB z (:67-81), negative python indices in the local search (:205-229).
"""
import math

import numpy as np
import scipy.signal
import sklearn.neighbors
from sklearn.preprocessing import MinMaxScaler, StandardScaler

from . import rfc
from .fits import line_best_fit
from .xform import inv_transform, transform_pts, unit_vector


def _pol2cart(arr):
    r, t = arr[:, 1], arr[:, 0]
    return np.c_[r * np.cos(t), r * np.sin(t)]


def _closest_angles(array, v):
    return np.abs([math.atan2(math.sin(v - a), math.cos(v - a)) for a in array])


def _peak_nearest(th):
    if len(th) == 1:
        return np.array([0])
    out = []
    for p in th:
        angs = _closest_angles(th, p)
        angs = angs[np.round(angs, 2) != 0]
        angs.sort()
        out.append(angs[0])
    return np.array(out)


def _peak_next_nearest(th):
    if len(th) == 1:
        return np.array([0])
    if len(th) == 2:
        return np.array([0, 0])
    out = []
    for p in th:
        angs = _closest_angles(th, p)
        angs = angs[np.round(angs, 2) != 0]
        angs.sort()
        out.append(angs[1])
    return np.array(out)


def x_process(polar, polar_0, zs, canal_axis, interp_num, n=7):
    """bicipital_groove.py:29-158 -> (X scaled, X raw, peak_theta, rows)."""
    z_scale = MinMaxScaler().fit_transform(zs.reshape(-1, 1)).flatten()
    canal_u = unit_vector(canal_axis[0], canal_axis[1])
    cols = [[] for _ in range(9)]
    peak_theta, rows = [], []
    for i, (rpol, rpol0) in enumerate(zip(polar, polar_0)):
        theta = rpol0[0]
        radius_og = rpol[1]
        radius = scipy.signal.savgol_filter(-1 * rpol0[1], 10, 1)
        rmin = -1 * np.argmin(radius)
        radius_roll = np.roll(radius, rmin)
        peaks, prop = scipy.signal.find_peaks(radius_roll, height=-10, prominence=0.6, width=0.1)
        peaks = (peaks - rmin) % interp_num
        if len(peaks) > n:
            part = np.sort(np.argpartition(prop["prominences"], -n)[-n:])   # B-5: ascending index
            peaks = peaks[part]
            prop = {k: np.asarray(v)[part] for k, v in prop.items()}
        th = theta[peaks]
        # canal_dist (:67-81)
        z_peaks = np.repeat(zs[i], len(th))
        canal_pts = (canal_u.reshape(-1, 1) @ z_peaks.reshape(1, -1))[:2, :]
        peak_pts = _pol2cart(np.c_[th, radius_og[peaks]]).T
        dist = np.sqrt(np.sum((peak_pts - canal_pts) ** 2, axis=0))
        cols[0].extend(radius_og[peaks])
        cols[1].extend(_peak_nearest(th))
        cols[2].extend(_peak_next_nearest(th))
        cols[3].extend([z_scale[i]] * len(peaks))
        cols[4].extend(prop["prominences"])
        cols[5].extend(prop["widths"])
        cols[6].extend(prop["width_heights"])
        cols[7].extend(dist)
        cols[8].extend(np.repeat(len(peaks) / n, len(peaks)))
        peak_theta.extend(th)
        rows.extend([i] * len(peaks))
    X_raw = np.c_[tuple(cols)]
    X = StandardScaler().fit_transform(X_raw)
    return X, X_raw, np.array(peak_theta), np.array(rows)


def find_nearest_idx(array, value):
    """bicipital_groove.py:287-293."""
    idx = np.searchsorted(array, value, side="left")
    return idx - 1 if idx == len(array) else idx


KDE_TIE = 1e-9


def kde_argmax_theta(theta_sel):
    """bicipital_groove.py:184-188 (sklearn KernelDensity(kernel='linear'), bandwidth 1.0)."""
    kde = sklearn.neighbors.KernelDensity(kernel="linear")
    kde.fit(np.asarray(theta_sel).reshape(-1, 1))
    tlin = np.linspace(-1 * np.pi, np.pi, 1024).reshape(-1, 1)
    bg_prob = np.exp(kde.score_samples(tlin))
    # canonical rule B-8: a linear-kernel density is exactly flat wherever as many samples lie within the bandwidth on either
    # side, so a maximum can span many grid points and `np.argmax` then follows sklearn's rounding noise; ties within 1e-9
    # (relative) go to the lowest grid index (off a plateau neighbouring values differ by >= 2e-5)
    return tlin[int(np.nonzero(bg_prob >= bg_prob.max() * (1.0 - KDE_TIE))[0][0])][0]


def groove_points(polar, zs, centroids, canal_axis_ct, T_obb, tables, interp_num=512, deg_window=7):
    """bicipital_groove.py:160-242.
    polar = itr_centered_start(cutoff) (R,2,M); zs (R,); centroids (R,2).
    -> dict(points_obb, points_ct, bg_theta, X, X_raw, peak_theta, proba, local_idx)."""
    polar_0 = polar.copy()
    polar_0[:, 1, :] = np.apply_along_axis(lambda x: x - np.mean(x), axis=1, arr=polar[:, 1, :])
    X, X_raw, peak_theta, rows = x_process(polar, polar_0, zs, canal_axis_ct, interp_num)
    proba1 = rfc.predict_proba1(tables, X)
    sel = proba1 > np.float32(0.4)                                   # :185
    if not sel.any():
        raise ValueError("no bicipital-groove candidate peak with P > 0.4")
    bg_theta = kde_argmax_theta(peak_theta[sel])
    ivar = int(round(deg_window / (360 / interp_num)))               # :192
    if ivar < 1:
        ivar = 1
    bg_xy = np.zeros((len(zs), 2))
    local_idx = np.zeros(len(zs), dtype=np.int64)
    for i in range(len(zs)):
        esti = find_nearest_idx(polar_0[i, 0, :].flatten(), bg_theta)
        if ivar > esti:
            rng = np.concatenate((polar_0[i, :, (esti - ivar):], polar_0[i, :, :(esti + ivar)]), axis=1)
        else:
            rng = polar_0[i, :, (esti - ivar):(esti + ivar)]
        loc = int(np.argmin(rng[1, :])) + (esti - ivar)
        local_idx[i] = loc
        bg_xy[i, :] = _pol2cart(polar[i, :, loc].reshape(1, 2))
    pts_obb = np.c_[bg_xy, zs] + np.c_[centroids, np.zeros(len(centroids))]
    pts_ct = transform_pts(pts_obb, inv_transform(T_obb))
    return dict(points_obb=pts_obb, points_ct=pts_ct, bg_theta=float(bg_theta), X=X, X_raw=X_raw,
                peak_theta=peak_theta, rows=rows, proba=proba1, local_idx=local_idx)


def groove_axis(points_obb, T_obb):
    """bicipital_groove.py:244-265 (+ canonical B-4)."""
    z = points_obb[:, 2]
    z_dist = np.max(z) - np.min(z)
    c, d = line_best_fit(points_obb)
    if d[2] < 0:
        d = -d
    ends = np.array([c + d * (z_dist / 2), c - d * (z_dist / 2)])
    return transform_pts(ends, inv_transform(T_obb))
