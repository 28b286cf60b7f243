"""Randomized parity sweep (GPU box): ONE ragged batch mixing similarity copies of all four fixtures in shuffled order, run through
submit / collect with the hull overlap on (second run uses prepared hulls), device f32 path vs oracle per humerus.
    SEED0=11 NPER=8 python tools/parity_sweep_ragged.py > gpurun_out/sweep_ragged.log"""
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
NPER, seed0 = int(os.environ.get("NPER", "8")), int(os.environ.get("SEED0", "11"))
meshes = []
for bi, name in enumerate(["humerus_left", "humerus_right", "humerus_left_trab", "humerus_left_flipped"]):
    v, f = load_stl(f"tests/golden/bones/{name}.stl")
    T = synth.similarity_transforms(NPER, v, seed=seed0 + bi)
    meshes += [(name, synth.apply_similarity(T[i], v), f) for i in range(NPER)]
order = np.random.default_rng(seed0).permutation(len(meshes))
meshes = [meshes[i] for i in order]
e.upload([(mv, mf) for _, mv, mf in meshes])
e.set_overlap(True)
e.submit(_lib.STAGE_ALL); first = e.collect().copy()
e.submit(_lib.STAGE_ALL); e.submit(_lib.STAGE_ALL); second = e.collect().copy(); third = e.collect().copy()
print("runs identical:", first.tobytes() == second.tobytes() == third.tobytes(), flush=True)
worst = 0.0; nbad = 0
for i, (name, mv, mf) in enumerate(meshes):
    h = OracleHumerus(mv, mf, tables, W, unet_eval="chain"); L = h.landmarks(); r = third[i]
    ok = r["status"] == 0 and float(r["bg_theta"]) == L["bg_theta"] and int(r["n_anp"]) == len(L["anp_points"]) and bool(r["flipped"]) == h.obb["flipped"]
    d = max(float(np.abs(np.asarray(r[k]).reshape(np.shape(L[k])) - L[k]).max()) for k in ("canal_axis", "te_axis", "groove_axis", "anp_plane_point", "anp_axis_normal", "anp_axis_central", "csys"))
    worst = max(worst, d); nbad += 0 if ok else 1
    print(i, name, "OK" if ok else "MISMATCH", "max diff %.2e" % d, flush=True)
print("ragged batch of", len(meshes), ": mismatches", nbad, "worst diff %.2e mm" % worst)
