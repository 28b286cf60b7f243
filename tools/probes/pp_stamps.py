"""Where a ping-pong kernel's cycles go (library built with SHOULDER_HIPCC_FLAGS=-DPP_STAMP into another path: SHOULDER_LIB=...).
Runs the network alone a few times on 64 x 512 x 512 and prints, per part of a phase, the mean cycles per phase of the ON / OFF waves.
  SHOULDER_LIB=ab/libpp_stamp.so python tools/probes/pp_stamps.py"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from shoulder_amd import _lib, unet_spec
from shoulder_amd.engine import Engine

eng = Engine(0)
eng.load_unet(unet_spec.make_teacher_weights(), unet_spec.BASE, unet_spec.DEPTH)
eng.set_params(unet_dtype=_lib.UNET_BF16)
img = np.random.default_rng(1).random((64, 512, 512), dtype=np.float32)
for _ in range(3): eng.unet_infer(img)
L = ctypes.CDLL(os.environ["SHOULDER_LIB"])
NS = 8
out = (ctypes.c_ulonglong * (3 * 256 * 8 * NS))()
rc = L.sh_lab_pp_stamps(out)
both = np.frombuffer(out, dtype=np.uint64).reshape(3, 256, 8, NS).astype(np.float64)
which = int(os.environ.get("PP_KERNEL", "0"))      # 0 dec0b + head, 1 enc0 (its parts: see k_enc0_pp)
a = both[which]
if which == 1: names_override = ["ON multiply", "OFF 1st conv", "ON barrier", "OFF patch ld", "OFF store+epi", "OFF barrier", "phases", "whole loop"]
names = ["ON multiply", "ON vm wait", "ON barrier", "OFF stage", "OFF epilogue", "OFF barrier", "phases", "whole loop"]
if which == 1: names = names_override
if which == 2: names = ["ON multiply", "OFF up-conv", "ON barrier", "OFF pieces", "OFF epi+vmwait", "OFF barrier", "phases", "whole loop"]
print("rc", rc, "phases per wave: mean", a[:, :, 6].mean(), "loop cycles mean", a[:, :, 7].mean(), "max", a[:, :, 7].max())
ph = a[:, :, 6].mean()
for i in range(6):
    print(f"{names[i]:14s} cycles per phase-of-that-kind: group0 {2 * a[:, :4, i].mean() / ph:8.0f}  group1 {2 * a[:, 4:, i].mean() / ph:8.0f}")
print("cycles per phase (loop / phases):", a[:, :, 7].mean() / ph)
print("per wave (mean over workgroups), cycles per own phase:")
for w in range(8):
    print(f" wave {w}: " + "  ".join(f"{names[i][:12]}={2 * a[:, w, i].mean() / ph:6.0f}" for i in range(6)))
