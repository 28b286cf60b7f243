"""shoulder_amd.stl.load_stl (host loader of the facade; `trimesh.load_mesh` at mesh.py:22-27 reads both STL flavours): an ASCII
file gives the mesh its binary twin gives; malformed files are ValueErrors."""
import os
import struct

import numpy as np
import pytest

from conftest import BONES
from shoulder_amd.stl import load_stl


def test_ascii_equals_binary(tmp_path):
    v, f = load_stl(os.path.join(BONES, "humerus_right.stl"))
    tris = f[:3000]
    a = tmp_path / "a.stl"
    with open(a, "w") as fh:
        fh.write("solid bone\n")
        for t in tris:
            fh.write(" facet normal 0 0 0\n  outer loop\n")
            for i in t:
                fh.write("   vertex %r %r %r\n" % tuple(float(x) for x in v[i]))
            fh.write("  endloop\n endfacet\n")
        fh.write("endsolid bone\n")
    b = tmp_path / "b.stl"
    with open(b, "wb") as fh:
        fh.write(b" " * 80 + struct.pack("<I", len(tris)))
        for t in tris:
            fh.write(struct.pack("<12fH", 0, 0, 0, *v[t[0]], *v[t[1]], *v[t[2]], 0))
    av, af = load_stl(a)
    bv, bf = load_stl(b)
    assert np.array_equal(av.view(np.uint32), bv.view(np.uint32)) and np.array_equal(af, bf)


@pytest.mark.parametrize("blob", [b"", b"x" * 50, b"x" * 84, b"solid empty\nendsolid empty\n", b"solid t\n vertex 0 0 0\n vertex 1 0 0\nendsolid\n",
                                  b"solid t\n vertex 0 0 0\n vertex 1 0 nan\n vertex 0 1 0\nendsolid\n"])
def test_malformed_files(tmp_path, blob):
    p = tmp_path / "bad.stl"
    p.write_bytes(blob)
    with pytest.raises(ValueError):
        load_stl(p)
