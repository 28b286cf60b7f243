"""Parity check (GPU box) on meshes with more than one connected component: the humerus plus a detached fragment beside the shaft,
and the humerus with a closed cavity inside the head (inner loops in the sections).  Device f32 path vs oracle.
    python tools/parity_sweep_fragments.py > gpurun_out/sweep_frag.log"""
import sys, os, numpy as np, subprocess
sys.path.insert(0, os.getcwd())
from shoulder_amd import _lib, unet_spec
from shoulder_amd.engine import Engine
from shoulder_amd.stl import load_stl
from oracle.humerus import OracleHumerus
from oracle import rfc
subprocess.run(["make", "-C", "oracle"], capture_output=True)
tables = rfc.load_tables("shoulder_amd/models/rfc_bg3.npz")
W = unet_spec.make_teacher_weights()
e = Engine(0); e.load_rfc(); e.load_unet(W, 32, 4); e.set_params(unet_dtype=_lib.UNET_F32)
v, f = load_stl("tests/golden/bones/humerus_left.stl")
h0 = OracleHumerus(v, f, tables, W, unet_eval="chain")
Tinv = np.linalg.inv(h0.T_obb)


def box(center_obb, half, inward=False):
    c = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], dtype=np.float64) * half + center_obb
    q = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    t = []
    for a, b, cc, d in q: t += [(a, b, cc), (a, cc, d)]
    t = np.array(t, dtype=np.int32)
    if inward: t = t[:, ::-1]
    ct = (np.c_[c, np.ones(8)] @ Tinv.T)[:, :3]
    return ct.astype(np.float32), t


zmax = h0.verts_obb[:, 2].max()
cases = {"fragment beside the shaft": box(np.array([45.0, 3.0, -20.0]), np.array([2.5, 2.0, 3.0])),
         "cavity inside the head": box(np.array([0.0, 0.0, zmax - 22.0]), np.array([3.0, 3.5, 4.0]), inward=True)}
for tag, (bv, bf) in cases.items():
    mv = np.concatenate([v, bv]); mf = np.concatenate([f, bf + len(v)]).astype(np.int32)
    try:
        e.upload([(mv, mf)]); r = e.run(_lib.STAGE_ALL)[0]; dev = None
    except Exception as ex:
        dev = str(ex)[:160]
    try:
        h = OracleHumerus(mv, mf, tables, W, unet_eval="chain"); L = h.landmarks(); orc = None
    except Exception as ex:
        orc = type(ex).__name__ + ": " + str(ex)[:160]
    if dev or orc:
        print(tag, "| device:", dev or "ok", "| oracle:", orc or "ok", flush=True); continue
    ok = r["status"] == 0 and float(r["bg_theta"]) == L["bg_theta"] and int(r["n_anp"]) == len(L["anp_points"]) and bool(r["flipped"]) == h.obb["flipped"]
    d = max(float(np.abs(np.asarray(r[k]).reshape(np.shape(L[k])) - L[k]).max()) for k in ("canal_axis", "te_axis", "groove_axis", "anp_plane_point", "anp_axis_normal", "anp_axis_central", "csys"))
    print(tag, "OK" if ok else "MISMATCH", "max diff %.2e mm" % d, "n_anp", int(r["n_anp"]), len(L["anp_points"]), flush=True)
