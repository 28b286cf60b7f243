"""K1: binary STL ingest + vertex merge (oracle; test infrastructure).

Follows `trimesh.load_mesh(..., process=True)` as called at reference
`src/shoulder/humerus/mesh.py:22-27` (trimesh 3.23.5, poetry.lock:4145):
binary STL record = (normal 3xf4, vertices 3x3xf4, attr u2); vertices that are
exactly equal merge into one.  Canonical deviation: merged vertex
order is order of first appearance in the file (trimesh's is hash-sorted); faces
that reference the same vertex twice are dropped.
"""
import numpy as np

_REC = np.dtype([("n", "<f4", (3,)), ("v", "<f4", (3, 3)), ("a", "<u2")])


def read_stl_triangles(path) -> np.ndarray:
    """-> (F,3,3) float32 triangle soup."""
    raw = np.fromfile(str(path), dtype=np.uint8)
    if raw.size < 84:
        raise ValueError(f"{path}: not a binary STL (too short)")
    n = int(np.frombuffer(raw[80:84].tobytes(), dtype="<u4")[0])
    if raw.size != 84 + 50 * n:
        raise ValueError(f"{path}: not a binary STL (size {raw.size} != 84+50*{n})")
    rec = np.frombuffer(raw[84:].tobytes(), dtype=_REC)
    return np.ascontiguousarray(rec["v"])


def merge_vertices(tris: np.ndarray):
    """(F,3,3) f32 -> verts (V,3) f32 in first-appearance order, faces (F',3) int32."""
    flat = tris.reshape(-1, 3)
    # exact-equality merge on the bit patterns (-0.0 == 0.0 is normalised first)
    flat = flat + np.float32(0.0)
    key = np.ascontiguousarray(flat).view(np.dtype((np.void, 12))).ravel()
    _, first, inv = np.unique(key, return_index=True, return_inverse=True)
    # re-number unique ids by first appearance
    order = np.argsort(first, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    verts = flat[first[order]]
    faces = rank[inv].reshape(-1, 3).astype(np.int32)
    ok = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])
    return np.ascontiguousarray(verts), np.ascontiguousarray(faces[ok])


def load_stl(path):
    return merge_vertices(read_stl_triangles(path))
