"""TEST INFRASTRUCTURE (oracle): the coordinate-system bookkeeping of the reference facade, restated.

Follows /root/reference/src/shoulder/bone.py:66-105 (`apply_csys_obb`, `apply_csys_ct`, `apply_csys_custom`, `apply_translation`),
base.py:24-63 (`_update_landmark_data`, `Transform`) and the `transform_landmark` methods of the landmark classes
(canal.py:126-130, bicipital_groove.py:267-271, anatomic_neck.py:238-248, epicondyle.py:103-105, surgical_neck.py:76-80): every
landmark is cached in CT and re-expressed through the ONE shared matrix.  Reference quirks kept: `apply_csys_custom(from_ct=False)`
and `apply_translation` left-multiply the matrix in force and then apply that CUMULATIVE matrix to the mesh AS IT STANDS
(bone.py:92-94, :100-102), so the mesh is not `T_total * mesh_ct` after them.  Pinned by tests/golden/csys_golden.npz
(generated from the reference's own code by tests/golden/make_csys_golden.py).  Only tests import this module."""
import numpy as np

from .xform import transform_pts, translate_transform


class CsysState:
    def __init__(self, verts_ct, landmarks_ct):
        """landmarks_ct: name -> (n, 3) array in CT."""
        self.matrix = np.identity(4)
        self.verts_ct = np.array(verts_ct, dtype=np.float64)
        self.mesh = self.verts_ct.copy()
        self.lm_ct = {k: np.array(v, dtype=np.float64) for k, v in landmarks_ct.items()}
        self.lm = {k: v.copy() for k, v in self.lm_ct.items()}

    def _set(self, M):
        M = np.asarray(M)
        if not isinstance(M, np.ndarray) or M.shape != (4, 4):      # base.py:55-58
            raise ValueError("Invalid transformation matrix shape")
        self.matrix = M
        self.lm = {k: transform_pts(v, self.matrix) for k, v in self.lm_ct.items()}      # base.py:32-35

    def apply_csys_ct(self):                     # bone.py:75-82
        self._set(np.identity(4))
        self.mesh = self.verts_ct.copy()
        return self.matrix

    def apply_csys_custom(self, T, from_ct=True):      # bone.py:84-95
        if from_ct:
            self._set(T)
            self.mesh = transform_pts(self.verts_ct, self.matrix)
        else:
            self._set(np.dot(T, self.matrix))
            self.mesh = transform_pts(self.mesh, self.matrix)
        return self.matrix

    def apply_translation(self, t):              # bone.py:97-105
        self._set(np.dot(translate_transform(np.asarray(t, dtype=np.float64)), self.matrix))
        self.mesh = transform_pts(self.mesh, self.matrix)
        return self.matrix
