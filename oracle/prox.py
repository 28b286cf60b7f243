"""`shoulder.ProximalHumerus` (cut humeri), CPU (oracle; test infrastructure).

Restates reference `src/shoulder/humerus/mesh.py:128-192` (`ProxObb._obb`) and the proximal variants of the
landmark classes wired by `src/shoulder/bone.py:24-64`:
  * `ProxObb._obb`: `apply_obb` (B-3 frame, no circle fits), 100 sections between 0.99*zmin and 0.99*zmax,
    `Path2D.area` of each (= outer loops minus holes; here |sum of the signed loop areas|, loops in canonical order),
    head = largest area -> flip so that it is at +z, then the canal range = longest run of consecutive sections
    with `np.gradient(savgol_filter(area, 3, 1)) < 10` -> `cutoff_pcts`;
  * `SurgicalNeck(only_proximal=True)`: change point on `areas1((0.2, 0.99))` (surgical_neck.py:25-26);
  * `Canal(proximal=True)`: default cutoffs replaced by `obb.cutoff_pcts` (canal.py:33-38);
  * `apply_csys_canal_articular`: `construct_csys(canal axis, anatomic-neck normal axis)` (bone.py:53-62).
Parity UNPINNED like the rest of the trimesh-backed code; the reference ships no cut-humerus fixture, the test mesh is
cut from `humerus_left.stl` by `tests/golden/make_proximal_fixture.py`.
"""
import numpy as np
import scipy.signal

from . import canal, cpd, obb
from .humerus import OracleHumerus
from .section import ZSlicer
from .slices import cutoff_range
from .xform import construct_csys, inv_transform, transform_pts

NUM_ZS = 100


def total_area(slicer: ZSlicer, z: float) -> float:
    """`Path2D.area` of one section: |sum of the signed shoelace areas| of its closed loops, loops ordered by their
    canonical start key, each ring walked from that start in segment order (the order the product sums in)."""
    sk, ek, sp, _ = slicer.segments(z)
    n = len(sk)
    if n == 0:
        return 0.0
    idx_of_start = {int(k): i for i, k in enumerate(sk)}
    nxt = np.array([idx_of_start.get(int(k), -1) for k in ek], dtype=np.int64)
    seen = np.zeros(n, dtype=bool)
    loops = []
    for i0 in range(n):
        if seen[i0]:
            continue
        chain, i, closed = [], i0, False
        while i >= 0 and not seen[i]:
            seen[i] = True
            chain.append(i)
            i = nxt[i]
            if i == i0:
                closed = True
                break
        if not closed or len(chain) < 3:
            continue
        chain = np.array(chain)
        k0 = int(np.argmin(sk[chain]))
        chain = np.r_[chain[k0:], chain[:k0]]
        loops.append((int(sk[chain[0]]), sp[chain]))
    loops.sort(key=lambda t: t[0])
    tot = 0.0
    for _, p in loops:
        a2 = 0.0
        for q in range(len(p)):
            qn = 0 if q + 1 == len(p) else q + 1
            a2 += p[q, 0] * p[qn, 1] - p[qn, 0] * p[q, 1]
        tot += 0.5 * a2
    return abs(tot)


def consecutive(arr):
    """mesh.py:139-140: longest run of consecutive integers (first of the longest)."""
    return max(np.split(arr, (np.where(np.diff(arr) != 1)[0] + 1)), key=len)


def canal_range(z_area):
    """mesh.py:181-190 -> (canal_zs indices, cutoff_pcts)."""
    grad = np.gradient(scipy.signal.savgol_filter(np.asarray(z_area, dtype=np.float64), 3, 1))
    canal_zs = consecutive(np.where(grad < 10)[0])
    return canal_zs, [canal_zs[0] / NUM_ZS, canal_zs[-1] / NUM_ZS], grad


def prox_obb(verts: np.ndarray, faces: np.ndarray):
    """mesh.py:134-192 -> dict(transform, z_bounds, z_length, verts_obb, flipped, z_area, cutoff_pcts, cutoff_bot)."""
    T_pre, ext, vol = obb.oriented_bounds(verts)
    v = transform_pts(verts, T_pre)
    z_bounds = (float(v[:, 2].min()), float(v[:, 2].max()))
    z_length = abs(z_bounds[0]) + abs(z_bounds[1])
    z_intervals = np.linspace(z_bounds[0] * 0.99, z_bounds[1] * 0.99, NUM_ZS).flatten()
    sl = ZSlicer(v, faces)
    z_area = [total_area(sl, float(z)) for z in z_intervals]
    humeral_head_z = z_intervals[int(np.argmax(z_area))]
    flipped = bool(humeral_head_z < 0)
    flip = obb.FLIP if flipped else np.identity(4)
    if flipped:
        v = transform_pts(v, flip)
        z_area = z_area[::-1]
    canal_zs, cutoff_pcts, grad = canal_range(z_area)
    return dict(transform=np.matmul(flip, T_pre), z_bounds=z_bounds, z_length=z_length, verts_obb=v, flipped=flipped,
                z_area=np.asarray(z_area), grad=grad, cutoff_pcts=cutoff_pcts, cutoff_bot=int(canal_zs[0]),
                canal_zs=(int(canal_zs[0]), int(canal_zs[-1])), extents=ext, volume=vol)


class OracleProximalHumerus(OracleHumerus):
    """bone.py:24-64."""

    @property
    def obb(self):
        return self._memo("obb", lambda: prox_obb(self.verts.astype(np.float64), self.faces))

    @property
    def neck(self):
        def f():
            cutoff = (0.2, 0.99)                                   # surgical_neck.py:25-26
            areas = self.full.cut(self.full.areas1_all, cutoff)
            bkp = cpd.kernel_cpd_one_bkp(areas)
            neck_z = float(self.full.zs(cutoff)[bkp])
            rings = ZSlicer(self.verts_obb, self.faces).loops(neck_z)
            if len(rings) > 1:
                ring = rings[int(np.argmin([np.sum(np.abs(np.mean(r[:, :2], axis=0))) for r in rings]))]
            else:
                ring = rings[0]
            pts_obb = np.c_[ring, np.full(len(ring), neck_z)]
            return dict(neck_z=neck_z, bkp=int(bkp), areas=areas, points_obb=pts_obb,
                        points_ct=transform_pts(pts_obb, inv_transform(self.T_obb)))
        return self._memo("neck", f)

    @property
    def canal(self):
        def f():
            cut = tuple(self.obb["cutoff_pcts"])                   # canal.py:33-38
            p_obb, p_ct = canal.canal_points(self.full.centroids_all, self.full.zs_all, self.T_obb, cut)
            a_obb, a_ct = canal.canal_axis(p_obb, self.obb["z_length"], self.T_obb, cut)
            return dict(points_obb=p_obb, points_ct=p_ct, axis_obb=a_obb, axis_ct=a_ct, cutoff=cut)
        return self._memo("canal", f)

    @property
    def te(self):
        raise AttributeError("a proximal humerus has no epicondyles (bone.py:24-64)")

    def csys_canal_articular(self):
        return construct_csys(self.canal["axis_ct"], self.anp["axis_normal_ct"])      # bone.py:57-59

    def metrics(self, axis_normal_current=None):
        from . import metrics as m
        side = m.side(self.canal["axis_ct"], self.anp["axis_central_ct"], self.groove["points_ct"])
        return dict(side=side, neckshaft=m.neckshaft(self.canal["axis_ct"], self.anp["axis_normal_ct"]),
                    radius_curvature=m.spherefit(self.anp["articular_obb"])[0])

    def landmarks(self, with_unet=True):
        out = dict(T_obb=self.T_obb, z_length=self.obb["z_length"], neck_z=self.neck["neck_z"],
                   canal_axis=self.canal["axis_ct"], canal_points=self.canal["points_ct"],
                   groove_points=self.groove["points_ct"], groove_axis=self.groove["axis_ct"],
                   bg_theta=self.groove["bg_theta"], groove_local_idx=self.groove["local_idx"],
                   cutoff_pcts=self.obb["cutoff_pcts"])
        if with_unet:
            out.update(anp_points=self.anp["points_ct"], anp_plane_point=self.anp["plane_point_ct"],
                       anp_plane_normal=self.anp["plane_normal_ct"], anp_axis_normal=self.anp["axis_normal_ct"],
                       anp_axis_central=self.anp["axis_central_ct"], csys=self.csys_canal_articular())
        return out
