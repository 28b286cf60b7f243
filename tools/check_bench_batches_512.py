"""All 512 synthetic humeri of BASELINE configs[3] (8 shards of 64, seed 1234) through the bf16 path on one GPU: every status must be 0."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from shoulder_amd import _lib, synth, unet_spec
from shoulder_amd.engine import Engine
from shoulder_amd.stl import load_stl
verts, faces = load_stl("tests/golden/bones/humerus_left.stl")
e = Engine(0); e.load_rfc(); e.load_unet(unet_spec.make_teacher_weights(), 32, 4); e.set_params(unet_dtype=_lib.UNET_BF16)
e.upload([(verts, faces)])
bad = 0
ref = None
for r in range(8):
    T = synth.similarity_transforms(64, verts, seed=1234, start=r * 64)
    e.synth_batch(T)
    try:
        lm = e.run(_lib.STAGE_ALL)
    except Exception as ex:
        print("rank-shard", r, "FAILED:", str(ex)[:200]); bad += 1; continue
    nb = int((lm["status"] != 0).sum())
    print("rank-shard", r, "bad meshes", nb, "neckshaft range", float(lm["neckshaft"].min()), float(lm["neckshaft"].max()), "sides", np.bincount(lm["side"], minlength=2).tolist(), flush=True)
    bad += nb
print("TOTAL BAD", bad)
