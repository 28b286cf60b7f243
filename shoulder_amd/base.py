"""Shared facade types (mirror of reference src/shoulder/base.py + the third-party return types
the reference hands out: `trimesh.Trimesh` as `.mesh`, `skspatial.objects.Plane` from `.plane()`)."""
from abc import ABC, abstractmethod

import numpy as np


class Landmark(ABC):
    """base.py:9-16"""

    def _graph_obj(self):
        return None     # plotting (plotting.py) is out of scope of the hot path

    @abstractmethod
    def transform_landmark(self) -> None:
        """re-express an already computed landmark in the current coordinate system"""


class Transform:
    """base.py:45-63: shared, mutable 4x4 with shape validation."""

    def __init__(self, matrix=None):
        self._matrix = np.identity(4) if matrix is None else matrix

    @property
    def matrix(self) -> np.ndarray:
        return self._matrix

    @matrix.setter
    def matrix(self, new_matrix):
        if not isinstance(new_matrix, np.ndarray) or new_matrix.shape != (4, 4):
            raise ValueError("Invalid transformation matrix shape")
        self._matrix = new_matrix

    def reset(self):
        self._matrix = np.identity(4)


class Plane:
    """Stand-in for skspatial.objects.Plane (anatomic_neck.py:146-153): `.point`, `.normal`."""

    def __init__(self, point, normal):
        self.point = np.array(point, dtype=np.float64).reshape(3)
        self.normal = np.array(normal, dtype=np.float64).reshape(3)

    def __repr__(self):
        return f"Plane(point={self.point!r}, normal={self.normal!r})"


class Mesh:
    """Stand-in for the parts of trimesh.Trimesh user code touches (base.py:21, bone.py:62,155):
    `.vertices` (V,3) float64, `.faces` (F,3) int, `.bounds`, `.copy()`, `.apply_transform(T)`."""

    def __init__(self, vertices, faces, _engine=None):
        self.vertices = np.array(vertices, dtype=np.float64).reshape(-1, 3)
        self.faces = np.asarray(faces).reshape(-1, 3)
        self._engine = _engine

    @property
    def bounds(self):
        return np.stack([self.vertices.min(axis=0), self.vertices.max(axis=0)])

    def copy(self):
        return Mesh(self.vertices.copy(), self.faces.copy(), self._engine)

    def apply_transform(self, T):
        T = np.asarray(T, dtype=np.float64)
        if T.shape != (4, 4):
            raise ValueError("Invalid transformation matrix shape")
        if self._engine is None:
            raise RuntimeError("Mesh.apply_transform needs the HIP engine (no CPU fallback)")
        self.vertices = self._engine.transform_points(self.vertices, T)     # k_affine_f64 on the device
        return self


class Bone(ABC):
    """base.py:24-42"""

    def _list_landmarks(self):
        out = []
        for name in dir(self):          # alphabetical, as the reference (base.py:26)
            if name.startswith("__"):
                continue
            attr = getattr(self, name, None)
            if isinstance(attr, Landmark):
                out.append(attr)
        return out

    def _update_landmark_data(self):
        for land in self._list_landmarks():
            land.transform_landmark()
