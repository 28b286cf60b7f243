"""Cut a proximal-humerus test mesh out of the reference's `humerus_left.stl` (the reference ships no cut-humerus fixture;
`shoulder.ProximalHumerus`, bone.py:24-64, is meant for CT scans that end in the shaft).

    PYTHONPATH=/root/repo python tests/golden/make_proximal_fixture.py

The bone is clipped by a plane perpendicular to its long (OBB z) axis, `KEEP` of its length from the head end, and the
cut is closed by a fan of triangles around the section's centroid, so the result is a watertight, outward-oriented mesh
again.  Output: tests/golden/bones/proximal_left_cut.stl (binary STL, float32, CT coordinates of the original file).
"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import obb as o_obb                      # noqa: E402
from oracle.stl import load_stl                      # noqa: E402
from oracle.xform import inv_transform, transform_pts   # noqa: E402

KEEP = 0.58


def cut_above(v, f, zc):
    """Part of the closed mesh (v, f) with z >= zc, capped.  -> (verts, faces)."""
    z = v[:, 2]
    assert np.abs(z - zc).min() > 1e-7
    above = z > zc
    new_id, verts, faces = {}, [p for p in v], []

    def cross(a, b):          # vertex on edge a-b (a above, b below), shared between the two triangles of the edge
        key = (min(a, b), max(a, b))
        if key not in new_id:
            lo, hi = key
            t = (zc - v[lo, 2]) / (v[hi, 2] - v[lo, 2])
            p = v[lo] + t * (v[hi] - v[lo])
            p[2] = zc
            new_id[key] = len(verts)
            verts.append(p)
        return new_id[key]

    cut_edges = []            # directed along the kept surface's boundary
    for tri in f:
        s = above[tri]
        k = int(s.sum())
        if k == 3:
            faces.append(tuple(tri))
        elif k == 1:
            i = int(np.argmax(s))
            a, b, c = tri[i], tri[(i + 1) % 3], tri[(i + 2) % 3]
            pb, pc = cross(a, b), cross(a, c)
            faces.append((a, pb, pc))
            cut_edges.append((pb, pc))
        elif k == 2:
            i = int(np.argmin(s))
            a, b, c = tri[i], tri[(i + 1) % 3], tri[(i + 2) % 3]      # a below
            pb, pc = cross(b, a), cross(c, a)
            faces.append((pb, b, c))
            faces.append((pb, c, pc))
            cut_edges.append((pc, pb))
    # boundary loop(s) of the kept part; the cap's triangles run against them
    nxt = {a: b for a, b in cut_edges}
    assert len(nxt) == len(cut_edges)
    verts = np.array(verts)
    seen = set()
    for start in list(nxt):
        if start in seen:
            continue
        loop, i = [], start
        while i not in seen:
            seen.add(i)
            loop.append(i)
            i = nxt[i]
        assert i == start and len(loop) >= 3
        centre = len(verts)
        verts = np.vstack([verts, verts[loop].mean(axis=0)])
        for q in range(len(loop)):
            faces.append((centre, loop[(q + 1) % len(loop)], loop[q]))
    faces = np.array(faces, dtype=np.int64)
    used = np.unique(faces)
    remap = -np.ones(len(verts), dtype=np.int64)
    remap[used] = np.arange(len(used))
    return verts[used], remap[faces]


def write_stl(path, v, f):
    v = v.astype(np.float32)
    with open(path, "wb") as fh:
        fh.write(b"proximal humerus cut from humerus_left.stl (make_proximal_fixture.py)".ljust(80, b" "))
        fh.write(struct.pack("<I", len(f)))
        for tri in f:
            a, b, c = v[tri[0]].astype(np.float64), v[tri[1]].astype(np.float64), v[tri[2]].astype(np.float64)
            n = np.cross(b - a, c - a)
            ln = np.linalg.norm(n)
            n = n / ln if ln > 0 else n
            fh.write(struct.pack("<12fH", *n.astype(np.float32), *v[tri[0]], *v[tri[1]], *v[tri[2]], 0))


def main():
    verts, faces = load_stl(os.path.join(HERE, "bones", "humerus_left.stl"))
    o = o_obb.full_obb(verts.astype(np.float64), faces)
    vo = o["verts_obb"]                                   # head at +z
    zmin, zmax = vo[:, 2].min(), vo[:, 2].max()
    zc = zmax - KEEP * (zmax - zmin)
    v2, f2 = cut_above(vo, faces.astype(np.int64), float(zc))
    # signed volume > 0: outward orientation survived
    a, b, c = v2[f2[:, 0]], v2[f2[:, 1]], v2[f2[:, 2]]
    vol = np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6.0
    assert vol > 0, vol
    v_ct = transform_pts(v2, inv_transform(o["transform"]))
    out = os.path.join(HERE, "bones", "proximal_left_cut.stl")
    write_stl(out, v_ct, f2)
    print(out, len(v2), "vertices", len(f2), "faces", "volume", round(float(vol), 1), "mm^3")


if __name__ == "__main__":
    main()
