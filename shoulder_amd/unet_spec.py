"""Anatomic-neck UNet: architecture and synthetic ("teacher") parameters.

The reference loads `humerus/models/unetcrf_anp.onnx`
(`src/shoulder/humerus/anatomic_neck.py:62-76`), a blob that is MISSING from the
reference tree (`.MISSING_LARGE_BLOBS:1-2`) -- neither architecture nor weights are
recoverable, only the I/O contract: float32[1,1,H,W] in [0,1] -> logits, mask =
logit > 0.  This module defines the builder's stand-in:

  4-level UNet, base 32 channels: per level two 3x3 conv (+folded BN bias) + ReLU,
  2x2 max-pool down, 2x2 stride-2 transposed conv up, concat [skip, up], 1x1 head.

`make_teacher_weights(seed)` produces seeded He-normal parameters plus a hand-set
"teacher" path: channel 0 carries the input image unchanged through
enc0 -> skip -> dec0 (identity taps), and the head reads it as
logit = gain*(x - tau) + eps*<random features>, so the mask is a thresholded radius
image with a network-shaped boundary.  It makes the downstream plane / ellipse /
ray-cast geometry well-posed without the real weights.  A user with the real model
replaces the dict (same keys/shapes) -- see INTEGRATION.md.
"""
import numpy as np

BASE = 32
DEPTH = 4


def channels(base=BASE, depth=DEPTH):
    return [base * (2 ** i) for i in range(depth + 1)]


def make_teacher_weights(seed=1234, base=BASE, depth=DEPTH, tau=0.6, gain=8.0, eps=0.25):
    rng = np.random.default_rng(seed)
    ch = channels(base, depth)
    w = {}

    def conv(name, cin, cout, k=3):
        std = np.sqrt(2.0 / (k * k * cin))
        w[name + "_w"] = (rng.standard_normal((k, k, cin, cout)) * std).astype(np.float32)
        w[name + "_b"] = (rng.standard_normal(cout) * 0.01).astype(np.float32)

    cin = 1
    for i in range(depth):
        conv(f"enc{i}a", cin, ch[i])
        conv(f"enc{i}b", ch[i], ch[i])
        cin = ch[i]
    conv("bota", ch[depth - 1], ch[depth])
    conv("botb", ch[depth], ch[depth])
    for i in reversed(range(depth)):
        conv(f"up{i}", ch[i + 1], ch[i], k=2)
        conv(f"dec{i}a", 2 * ch[i], ch[i])
        conv(f"dec{i}b", ch[i], ch[i])
    w["head_w"] = (rng.standard_normal(ch[0]) * (eps / np.sqrt(ch[0]))).astype(np.float32)
    # teacher path: output channel 0 of these layers = input channel 0 (centre tap 1)
    for name in ("enc0a", "enc0b", "dec0a", "dec0b"):
        w[name + "_w"][:, :, :, 0] = 0.0
        w[name + "_w"][1, 1, 0, 0] = 1.0
        w[name + "_b"][0] = 0.0
    w["head_w"][0] = np.float32(gain)
    w["head_b"] = np.float32(-gain * tau)
    return w


def n_params(w):
    return int(sum(np.asarray(v).size for v in w.values()))
