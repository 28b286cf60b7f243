"""The bench's own batch (device-generated similarity copies, seed 1234, first 64): device f32 path vs oracle on the host-generated copies."""
import sys, os, time, numpy as np, subprocess
sys.path.insert(0, os.getcwd())
from shoulder_amd import _lib, synth, unet_spec
from shoulder_amd.engine import Engine
from shoulder_amd.stl import load_stl
from oracle.humerus import OracleHumerus
from oracle import rfc
subprocess.run(["make", "-C", "oracle"], capture_output=True)
tables = rfc.load_tables("shoulder_amd/models/rfc_bg3.npz")
W = unet_spec.make_teacher_weights()
e = Engine(0); e.load_rfc(); e.load_unet(W, 32, 4); e.set_params(unet_dtype=_lib.UNET_F32)
v, f = load_stl("tests/golden/bones/humerus_left.stl")
B = 64
T = synth.similarity_transforms(B, v, seed=1234)
e.upload([(v, f)]); e.synth_batch(T)
lm = e.run(_lib.STAGE_ALL)
dv = e.fetch("verts", np.float32)[:B * len(v) * 3].reshape(B, len(v), 3)
worst = 0.0; nbad = 0
for i in range(B):
    mv = synth.apply_similarity(T[i], v)
    same = np.array_equal(mv, dv[i])
    h = OracleHumerus(mv, f, tables, W, unet_eval="chain"); L = h.landmarks(); r = lm[i]
    ok = same and r["status"] == 0 and float(r["bg_theta"]) == L["bg_theta"] and int(r["n_anp"]) == len(L["anp_points"]) and bool(r["flipped"]) == h.obb["flipped"]
    d = max(float(np.abs(np.asarray(r[k]).reshape(np.shape(L[k])) - L[k]).max()) for k in ("canal_axis", "te_axis", "groove_axis", "anp_plane_point", "anp_axis_normal", "anp_axis_central", "csys"))
    worst = max(worst, d); nbad += 0 if ok else 1
    print(i, "OK" if ok else "MISMATCH", "verts identical" if same else "VERTS DIFFER", "max diff %.2e" % d, flush=True)
print("bench batch: mismatches", nbad, "worst diff %.2e mm" % worst)
