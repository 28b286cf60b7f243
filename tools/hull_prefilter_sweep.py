import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from scipy.spatial import ConvexHull
from shoulder_amd import _lib, synth
from shoulder_amd.engine import Engine
from shoulder_amd.stl import load_stl
e = Engine(0)
tot = 0; worst_keep = 0
for bi, name in enumerate(["humerus_left", "humerus_right", "humerus_left_trab", "humerus_left_flipped", "proximal_left_cut"]):
    v, f = load_stl(f"tests/golden/bones/{name}.stl")
    B = 48
    e.upload([(v, f)]); e.synth_batch(synth.similarity_transforms(B, v, seed=4000 + bi))
    if name.startswith("prox"): e.set_params(bone_kind=_lib.BONE_PROXIMAL)
    lm = e.run(_lib.STAGE_OBB)
    V = len(v)
    koff = e.fetch("hullpre.koff", np.int64)[:B + 1]
    kept = e.fetch("hullpre.kept", np.float32).reshape(-1, 3)
    verts = e.fetch("verts", np.float32)[:B * V * 3].reshape(B, V, 3)
    npl = e.fetch("hullpre.npl", np.int32)[:B]
    for b in range(B):
        K = kept[koff[b]:koff[b + 1]]
        ks = set(map(tuple, K))
        P = verts[b]
        hull = ConvexHull(P.astype(np.float64)).vertices
        missing = [i for i in hull if tuple(P[i]) not in ks]
        assert not missing, (name, b, missing[:5])
        worst_keep = max(worst_keep, len(K) / V); tot += 1
    print(name, "ok", B, "copies; planes", int(npl.min()), int(npl.max()), "kept frac", float((np.diff(koff) / V).min()), float((np.diff(koff) / V).max()), flush=True)
print("all", tot, "hulls intact; worst kept fraction", worst_keep)
