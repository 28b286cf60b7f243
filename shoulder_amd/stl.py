"""Binary STL ingest for the facade (stands in for `trimesh.load_mesh(..., process=True)` at
reference src/shoulder/humerus/mesh.py:22-27): 84-byte header + 50-byte records; vertices that are
bit-for-bit equal are merged and numbered by first appearance in the file; triangles that use a
vertex twice are dropped.  Host-side I/O glue (SURVEY 8(f) #3 keeps on-device ingest as a next row)."""
import pathlib

import numpy as np


def load_stl(path):
    """-> (verts float32 (V,3), faces int32 (F,3))."""
    data = pathlib.Path(path).read_bytes()
    if len(data) < 84:
        raise ValueError(f"{path}: too short for a binary STL")
    ntri = int.from_bytes(data[80:84], "little")
    if len(data) != 84 + 50 * ntri:
        raise ValueError(f"{path}: not a binary STL ({len(data)} bytes for {ntri} triangles)")
    rec = np.frombuffer(data, dtype=np.uint8, offset=84).reshape(ntri, 50)
    corners = np.ascontiguousarray(rec[:, 12:48]).view("<f4").reshape(ntri * 3, 3)
    corners = corners + np.float32(0)                      # -0.0 -> +0.0 so equal values share one bit pattern
    keys = np.ascontiguousarray(corners).view("<u4").astype(np.uint64)
    order = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))
    sk = keys[order]
    new_group = np.ones(len(sk), dtype=bool)
    new_group[1:] = (sk[1:] != sk[:-1]).any(axis=1)
    gid_sorted = np.cumsum(new_group) - 1
    gid = np.empty(len(sk), dtype=np.int64)
    gid[order] = gid_sorted
    first = np.full(gid_sorted[-1] + 1, len(sk), dtype=np.int64)
    np.minimum.at(first, gid, np.arange(len(sk)))
    rank = np.empty(len(first), dtype=np.int64)
    rank[np.argsort(first, kind="stable")] = np.arange(len(first))
    faces = rank[gid].reshape(ntri, 3).astype(np.int32)
    verts = np.empty((len(first), 3), dtype=np.float32)
    verts[rank] = corners[first]
    keep = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])
    return verts, np.ascontiguousarray(faces[keep])
