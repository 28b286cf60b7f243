"""Kernel times of the slice / groove / TE stages per slice set (full, proximal, distal): python tools/time_slice_sets.py (on the GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from shoulder_amd import _lib, synth, unet_spec
from shoulder_amd.engine import Engine
from shoulder_amd.stl import load_stl
B = 64
verts, faces = load_stl(os.path.join(ROOT, "tests", "golden", "bones", "humerus_left.stl"))
eng = Engine(0); eng.load_rfc(); eng.load_unet(unet_spec.make_teacher_weights(), unet_spec.BASE, unet_spec.DEPTH)
eng.set_params(unet_dtype=_lib.UNET_BF16)
eng.upload([(verts, faces)]); eng.synth_batch(synth.similarity_transforms(B, verts, seed=1234))
try: eng.run(_lib.STAGE_ALL)
except Exception as e: print('run error', str(e)[:60])
for mask in (0x2, 0x10, 0x80, 0x20, 0x100):
    def r():
        try: eng.run(mask)
        except Exception: pass
    r(); eng.enable_timing(True); eng.reset_timers()
    for _ in range(5): r()
    print("mask", hex(mask))
    for k in ("k_make_planes", "k_slice_emit", "k_slice_link", "k_resample_polar", "k_groove_rows", "k_te_rows", "k_te_ends", "k_te_orient", "k_groove_kde", "k_groove_rfc", "k_groove_scale"):
        ms, n = eng.kernel_time_ms(k)
        if n: print("  %-20s %7.3f ms x %.1f" % (k, ms, n / 5))
    eng.enable_timing(False)
