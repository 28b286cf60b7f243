"""Fuzz of the device STL ingest (GPU box): 24 ragged files built from shuffled / duplicated / degenerate / -0.0 triangles, sh_upload_stl vs the host loader, bit for bit."""
import sys, os, struct, numpy as np
sys.path.insert(0, os.getcwd())
import bench
from shoulder_amd.engine import Engine
from shoulder_amd.stl import load_stl
import tempfile, pathlib
rng = np.random.default_rng(123)
v, f = load_stl("tests/golden/bones/humerus_left_trab.stl")
d = pathlib.Path(tempfile.mkdtemp())
paths = []
for i in range(24):
    n = int(rng.integers(200, 30000))
    idx = rng.choice(len(f), n, replace=rng.random() < 0.5)
    tri = v[f[idx]].copy()                                   # (n,3,3)
    k = int(rng.integers(0, 50))
    if k: tri[rng.integers(0, n, k), 1] = tri[rng.integers(0, n, k), 0]          # corners equal to another triangle's corner
    k = int(rng.integers(0, 20))
    if k:
        j = rng.integers(0, n, k); tri[j, 2] = tri[j, 1]                          # degenerate: a vertex used twice
    if rng.random() < 0.5: tri[rng.integers(0, n, 30)] *= np.float32(0.0) * np.float32(-1.0)   # -0.0 coordinates
    if rng.random() < 0.3: tri = tri[rng.permutation(n)]
    p = d / f"f{i}.stl"
    with open(p, "wb") as fh:
        fh.write(bytes(rng.integers(0, 255, 80, dtype=np.uint8))); fh.write(struct.pack("<I", n))
        rec = np.zeros((n, 50), np.uint8); rec[:, 12:48] = tri.reshape(n, 9).astype("<f4").view(np.uint8).reshape(n, 36); fh.write(rec.tobytes())
    paths.append(str(p))
e = Engine(0)
ok = True
try:
    e.upload_stl(paths)
except Exception as ex:
    print("upload_stl error:", str(ex)[:200]); ok = False
if ok:
    gv = e.fetch("verts", np.float32, (int(e.voff[-1]), 3)); gf = e.fetch("faces", np.int32, (int(e.foff[-1]), 3))
    bad = 0
    for b, p in enumerate(paths):
        try: hv, hf = load_stl(p)
        except Exception as ex: print(b, "host loader error", ex); bad += 1; continue
        dv, df = gv[e.voff[b]:e.voff[b + 1]], gf[e.foff[b]:e.foff[b + 1]]
        same = dv.shape == hv.shape and df.shape == hf.shape and np.array_equal(dv.view(np.uint32), hv.view(np.uint32)) and np.array_equal(df, hf)
        if not same: print(b, "MISMATCH", dv.shape, hv.shape, df.shape, hf.shape); bad += 1
    print("24 fuzzed files, mismatches:", bad)
