"""Subprocess of tests/test_gpu_end_to_end.py::test_lanes_that_yield_their_reserve: SHOULDER_CU_YIELD=1 is read once per process.
Two lanes (bf16 network: its ticketed launches then cover the whole chip and their last workgroups yield to the other lane's chain)
against one context alone: the same records."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from shoulder_amd import _lib, synth, unet_spec          # noqa: E402
from shoulder_amd.engine import Engine                    # noqa: E402
from shoulder_amd.stl import load_stl                     # noqa: E402
from conftest import BONES, _teacher_weights              # noqa: E402

verts, faces = load_stl(os.path.join(BONES, "humerus_left.stl"))
B = 48      # enough items for the 32-channel level's launches to cover every CU (yielding needs total >= CUs)
Ts = [synth.similarity_transforms(B, verts, seed=s) for s in (15, 16)]


def lane(T, turns):
    e = Engine(0)
    e.load_rfc()
    e.load_unet(_teacher_weights(), unet_spec.BASE, unet_spec.DEPTH)
    e.set_params(unet_dtype=_lib.UNET_BF16)
    e.upload([(verts, faces)])
    e.synth_batch(T)
    if turns:
        e.set_unet_turns(True)
        e.set_overlap(True)
    return e


refs = []
for T in Ts:
    e = lane(T, False)
    refs.append(e.run(_lib.STAGE_ALL).copy())
    e.close()
lanes = [lane(T, True) for T in Ts]
got, pend = [[], []], []
for s in range(8):
    if len(pend) >= 2:
        k = pend.pop(0)
        got[k].append(lanes[k].collect().copy())
    lanes[s % 2].submit(_lib.STAGE_ALL)
    pend.append(s % 2)
for k in pend:
    got[k].append(lanes[k].collect().copy())
ok = all((r["status"] == 0).all() for r in refs) and all(r.tobytes() == refs[k].tobytes() for k in range(2) for r in got[k]) and all(len(g) == 4 for g in got)
for e in lanes:
    e.close()
print("YIELD_OK" if ok else "YIELD_MISMATCH")
