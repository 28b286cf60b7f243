"""K5-K9: the slice layer (oracle; test infrastructure).

Restates reference `src/shoulder/humerus/slice.py`:
  Slices.__init__/_slices   slice.py:10-32   (z_orig = mean(zs); heights = zs - z_orig)
  _centroids                slice.py:34-39   (trimesh `Path2D.centroid` = AABB centre of all loops)
  _areas1                   slice.py:49-60   (area of the largest closed loop)
  _ixy/_resample_polygon    slice.py:65-80, :166-189
  _ixy_centered             slice.py:85-87
  _itr_start                slice.py:102-108
  _itr_centered_start       slice.py:136-144
  _cart2pol_no_sort         slice.py:200-206
  _cutoff                   slice.py:157-164
  FullSlices/ProximalSlices/DistalSlices._zs  slice.py:219-224, :248-253, :271-276
`resample_polygon`, `cart2pol_no_sort`, `cutoff_range`, `roll_to_argmin_theta` are
pinned by tests/golden/slice_golden.npz (captured from the reference's own code).
"""
import numpy as np

from .section import ZSlicer, ring_area


def cutoff_range(n: int, cutoff, return_odd=False):
    """slice.py:157-164 -> (start_i, end_i)."""
    start_i = int((1 - cutoff[1]) * n)
    end_i = int((1 - cutoff[0]) * n)
    if return_odd and ((len(range(n)[start_i:end_i]) % 2) == 0):
        end_i -= 1
    return start_i, end_i


def resample_polygon(xy: np.ndarray, interp_num: int) -> np.ndarray:
    """slice.py:166-189."""
    d = np.cumsum(np.r_[0, np.sqrt((np.diff(xy, axis=0) ** 2).sum(axis=1))])
    d_sampled = np.linspace(0, d.max(), interp_num)
    return np.c_[np.interp(d_sampled, d, xy[:, 0]), np.interp(d_sampled, d, xy[:, 1])]


def cart2pol_no_sort(x, y):
    """slice.py:200-206."""
    return np.vstack((np.arctan2(y, x), np.sqrt(x ** 2 + y ** 2)))


def roll_to_argmin_theta(pol):
    """slice.py:107 / :143: columns rotated so that column 0 is argmin(theta)."""
    k = int(np.argmin(pol[0]))
    return np.c_[pol[:, k:], pol[:, :k]]


class Slices:
    def __init__(self, verts_obb, faces, zs: np.ndarray, interp_num: int):
        self.zs_all = np.asarray(zs, dtype=np.float64)
        self.interp_num = interp_num
        self.z_orig = np.mean(self.zs_all)                      # slice.py:18
        self.z_incrs = self.zs_all - self.z_orig                # slice.py:19
        sl = ZSlicer(verts_obb, faces)
        # section_multiplane: plane k passes through origin + heights[k]*normal
        self.z_eff = self.z_orig + self.z_incrs
        self.loops = [sl.loops(float(z)) for z in self.z_eff]   # slice.py:26-28
        n = len(self.zs_all)
        self.centroids_all = np.zeros((n, 2))
        self.areas1_all = np.zeros(n)
        self.n_loops = np.zeros(n, dtype=np.int64)
        self.largest = []
        for i, rings in enumerate(self.loops):
            if not rings:
                raise ValueError(f"slice {i} (z={self.z_eff[i]}) does not intersect the mesh")
            allp = np.concatenate([r[:-1] for r in rings])
            self.centroids_all[i] = 0.5 * (allp.min(axis=0) + allp.max(axis=0))   # slice.py:38
            areas = [ring_area(r) for r in rings]
            j = int(np.argmax(areas))                            # slice.py:53-59
            self.areas1_all[i] = areas[j]
            self.n_loops[i] = len(rings)
            self.largest.append(rings[j])
        self._ixy = None

    # cached_property chain ---------------------------------------------------------
    @property
    def ixy_all(self):
        if self._ixy is None:
            self._ixy = np.stack([resample_polygon(r, self.interp_num).T for r in self.largest])
        return self._ixy

    @property
    def ixy_centered_all(self):
        return self.ixy_all - self.centroids_all[:, :, None]

    @property
    def itr_start_all(self):
        return np.stack([roll_to_argmin_theta(cart2pol_no_sort(p[0], p[1])) for p in self.ixy_all])

    @property
    def itr_centered_start_all(self):
        return np.stack([roll_to_argmin_theta(cart2pol_no_sort(p[0], p[1])) for p in self.ixy_centered_all])

    def cut(self, arr, cutoff):
        a, b = cutoff_range(len(arr), cutoff)
        return arr[a:b]

    def zs(self, cutoff):
        return self.cut(self.zs_all, cutoff)


def full_zs(z_max_bound, z_min_bound, n=200):
    """slice.py:219-224."""
    return np.linspace(0.99 * z_max_bound, 0.99 * z_min_bound, n)


def proximal_zs(z_max_bound, neck_z, n=600):
    """slice.py:248-253."""
    return np.linspace(0.99 * z_max_bound, neck_z, n)


def distal_zs(z_min_bound, n=200):
    """slice.py:271-276."""
    return np.linspace(0.99 * z_min_bound, 0, n)
