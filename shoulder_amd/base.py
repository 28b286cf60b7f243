"""Shared facade types (mirror of reference src/shoulder/base.py + the third-party return types
the reference hands out: `trimesh.Trimesh` as `.mesh`, `skspatial.objects.Plane` from `.plane()`)."""
from abc import ABC, abstractmethod

import numpy as np


class Landmark(ABC):
    """base.py:9-16"""

    def _graph_obj(self):
        """plotly trace(s) of the landmark, None while it has not been computed (base.py:10-12)"""
        return None

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

    def _need_engine(self, what):
        if self._engine is None:
            raise RuntimeError(f"Mesh.{what} needs the HIP engine (no CPU fallback)")

    def slice_plane(self, plane_origin, plane_normal):
        """The part of the mesh on the side `plane_normal` points to, uncapped (`trimesh.Trimesh.slice_plane` as called at
        arthroplasty.py:82-85): k_clip_* on the device."""
        self._need_engine("slice_plane")
        v, f = self._engine.slice_mesh_planes(self.vertices, self.faces, [plane_origin], [plane_normal])[0]
        return Mesh(v, f, self._engine)

    def section(self, plane_normal, plane_origin):
        """Cross-section with one plane (`trimesh.Trimesh.section` as called at arthroplasty.py:71): the cut edges of the
        same device pass, chained into closed loops."""
        self._need_engine("section")
        v, f, e = self._engine.slice_mesh_planes(self.vertices, self.faces, [plane_origin], [plane_normal], edges=True)[0]
        return Section(v, e, np.asarray(plane_normal, dtype=np.float64))


class _Polygon:
    def __init__(self, area):
        self.area = area


class Section:
    """Stand-in for the trimesh Path3D a section returns, as far as the reference touches it (arthroplasty.py:71-78):
    `.entities` (one per closed loop), `.discrete` (list of (k+1, 3) closed point loops), `.polygons_closed[i].area`.
    The start vertex and direction of a loop are implementation details of trimesh's graph traversal; here every loop
    starts at its smallest vertex id."""

    def __init__(self, vertices, edges, normal):
        adj = {}
        for a, b in np.asarray(edges).tolist():
            if a != b:
                adj.setdefault(a, []).append(b)
                adj.setdefault(b, []).append(a)
        loops, seen = [], set()
        for s in sorted(adj):
            if s in seen or len(adj[s]) != 2:
                continue
            loop, prev, cur, closed = [s], None, s, False
            seen.add(s)
            while True:
                cand = [x for x in adj[cur] if x != prev] or adj[cur]
                nx = cand[0]
                if nx == s:
                    closed = True
                    break
                if nx in seen or len(adj.get(nx, ())) != 2:
                    break
                loop.append(nx)
                seen.add(nx)
                prev, cur = cur, nx
            if closed and len(loop) >= 3:
                loops.append(loop)
        n = normal / np.linalg.norm(normal)
        u = np.cross(n, [1.0, 0.0, 0.0] if abs(n[0]) < 0.9 else [0.0, 1.0, 0.0])
        u /= np.linalg.norm(u)
        w = np.cross(n, u)
        self.entities = loops
        self.discrete = [np.asarray(vertices)[lp + lp[:1]] for lp in loops]
        areas = []
        for d in self.discrete:
            x, y = d @ u, d @ w
            areas.append(0.5 * abs(np.sum(x[:-1] * y[1:] - x[1:] * y[:-1])))
        self.polygons_closed = [_Polygon(a) for a in areas]


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

    def _list_landmarks_graph_obj(self):
        """graph objects of every computed landmark (base.py:37-42)"""
        objs = [lm._graph_obj() for lm in self._list_landmarks()]
        return [o for o in objs if o is not None]

    def _update_landmark_data(self):
        for land in self._list_landmarks():
            land.transform_landmark()
