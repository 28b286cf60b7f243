"""STL ingest for the facade (stands in for `trimesh.load_mesh(..., process=True)` at reference
src/shoulder/humerus/mesh.py:22-27).  Binary files: 84-byte header + 50-byte records.  ASCII files
(`solid ... facet normal ... vertex x y z ...`, which trimesh reads as well) are parsed to the same
float32 corner list.  Vertices that are bit-for-bit equal are merged and numbered by first
appearance in the file; triangles that use a vertex twice are dropped.  Host-side I/O for a single
file; batches of binary files are parsed and merged on the device (`sh_upload_stl`)."""
import pathlib
import re

import numpy as np

_VERTEX = re.compile(rb"vertex\s+(\S+)\s+(\S+)\s+(\S+)")


def _ascii_corners(path, data):
    """corner list of an ASCII STL, float32 like the binary form stores it"""
    xyz = _VERTEX.findall(data)
    if len(xyz) == 0 or len(xyz) % 3:
        raise ValueError(f"{path}: ASCII STL with {len(xyz)} vertex lines (not a multiple of 3)")
    try:
        return np.array(xyz, dtype=np.float64).astype(np.float32)
    except ValueError:
        raise ValueError(f"{path}: unreadable vertex coordinates in an ASCII STL") from None


def load_stl(path):
    """-> (verts float32 (V,3), faces int32 (F,3))."""
    data = pathlib.Path(path).read_bytes()
    ntri = int.from_bytes(data[80:84], "little") if len(data) >= 84 else -1
    if len(data) >= 84 and len(data) == 84 + 50 * ntri:
        rec = np.frombuffer(data, dtype=np.uint8, offset=84).reshape(ntri, 50)
        corners = np.ascontiguousarray(rec[:, 12:48]).view("<f4").reshape(ntri * 3, 3)
    elif data.lstrip()[:5].lower() == b"solid" and b"vertex" in data:
        corners = _ascii_corners(path, data)
        ntri = len(corners) // 3
    elif len(data) < 84:
        raise ValueError(f"{path}: too short for a binary STL")
    else:
        raise ValueError(f"{path}: not a binary STL ({len(data)} bytes for {ntri} triangles)")
    if not np.isfinite(corners).all():
        raise ValueError(f"{path}: NaN / infinite coordinates")
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
