"""K19: UNet forward (oracle; test infrastructure).

The reference's segmenter `unetcrf_anp.onnx` is a missing blob
(`.MISSING_LARGE_BLOBS:1-2`); only its I/O contract is recoverable
(`src/shoulder/humerus/anatomic_neck.py:62-85`: float32[1,1,H,W] in [0,1] -> logits
(H,W), mask = logit > 0).  Architecture here is builder-defined (DESIGN.md "UNet"):
4-level double-conv UNet, BN folded into conv (bias), ReLU, 2x2 max-pool, 2x2 stride-2
transposed-conv up-sampling, skip concat [skip, up], 1x1 head.  Parity for K19 is
therefore GPU-vs-this-restatement only ("parity unpinned" w.r.t. the reference).

Weights are DATA handed in by the caller as a dict:
  'enc{i}a_w' (3,3,Cin,C) 'enc{i}a_b' (C,)  'enc{i}b_w' 'enc{i}b_b'   i = 0..depth-1
  'bota_w' 'bota_b' 'botb_w' 'botb_b'
  'up{i}_w' (2,2,Cin,C) 'up{i}_b'  'dec{i}a_w' (3,3,2C,C) 'dec{i}a_b' 'dec{i}b_w' 'dec{i}b_b'
  'head_w' (C0,) 'head_b' ()
Two evaluators:
  forward_f64   -- NumPy float64 ("truth" for tolerance checks)
  forward_chain -- C restatement (oracle/unet_chain.c) that accumulates every output
                   as ONE float32 fma chain in (tap, cin) order from the bias, which is
                   what the f32 MFMA path of the product computes bit for bit.
"""
import ctypes
import os

import numpy as np


def depth_of(w):
    d = 0
    while f"enc{d}a_w" in w:
        d += 1
    return d


def _conv3(x, w, b, relu, dt):
    H, W, Cin = x.shape
    xp = np.zeros((H + 2, W + 2, Cin), dtype=dt)
    xp[1:-1, 1:-1] = x
    out = np.broadcast_to(b.astype(dt), (H * W, w.shape[3])).copy()
    for dy in range(3):
        for dx in range(3):
            out += xp[dy:dy + H, dx:dx + W].reshape(-1, Cin) @ w[dy, dx].astype(dt)
    out = out.reshape(H, W, -1)
    return np.maximum(out, 0) if relu else out


def _pool(x):
    H, W, C = x.shape
    return x.reshape(H // 2, 2, W // 2, 2, C).max(axis=(1, 3))


def _up(x, w, b, dt):
    H, W, Cin = x.shape
    C = w.shape[3]
    out = np.empty((H, 2, W, 2, C), dtype=dt)
    flat = x.reshape(-1, Cin)
    for dy in range(2):
        for dx in range(2):
            out[:, dy, :, dx, :] = (flat @ w[dy, dx].astype(dt) + b.astype(dt)).reshape(H, W, C)
    return out.reshape(2 * H, 2 * W, C)


def forward_f64(weights, image):
    """image (H,W) float -> logits (H,W) float64."""
    dt = np.float64
    d = depth_of(weights)
    x = np.asarray(image, dtype=np.float32).astype(dt)[:, :, None]
    skips = []
    for i in range(d):
        x = _conv3(x, weights[f"enc{i}a_w"], weights[f"enc{i}a_b"], True, dt)
        x = _conv3(x, weights[f"enc{i}b_w"], weights[f"enc{i}b_b"], True, dt)
        skips.append(x)
        x = _pool(x)
    x = _conv3(x, weights["bota_w"], weights["bota_b"], True, dt)
    x = _conv3(x, weights["botb_w"], weights["botb_b"], True, dt)
    for i in reversed(range(d)):
        x = _up(x, weights[f"up{i}_w"], weights[f"up{i}_b"], dt)
        x = np.concatenate([skips[i], x], axis=2)
        x = _conv3(x, weights[f"dec{i}a_w"], weights[f"dec{i}a_b"], True, dt)
        x = _conv3(x, weights[f"dec{i}b_w"], weights[f"dec{i}b_b"], True, dt)
    return x @ weights["head_w"].astype(dt) + dt(weights["head_b"])


def flops(weights, H, W):
    """2*MACs of every conv / up-conv / head at input size HxW."""
    d = depth_of(weights)
    total, h, w = 0, H, W
    for i in range(d):
        for k in ("a", "b"):
            s = weights[f"enc{i}{k}_w"].shape
            total += 2 * 9 * h * w * s[2] * s[3]
        h, w = h // 2, w // 2
    for k in ("a", "b"):
        s = weights[f"bot{k}_w"].shape
        total += 2 * 9 * h * w * s[2] * s[3]
    for i in reversed(range(d)):
        s = weights[f"up{i}_w"].shape
        total += 2 * 4 * h * w * s[2] * s[3]
        h, w = h * 2, w * 2
        for k in ("a", "b"):
            s = weights[f"dec{i}{k}_w"].shape
            total += 2 * 9 * h * w * s[2] * s[3]
    total += 2 * h * w * weights["head_w"].shape[0]
    return total


# ---- float32 fma-chain restatement (C) ---------------------------------------------
_LIB = None


def _chain_lib():
    global _LIB
    if _LIB is None:
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.path.join(here, "_build", "libunet_chain.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle`")
        lib = ctypes.CDLL(path)
        fp = ctypes.POINTER(ctypes.c_float)
        lib.oc_conv3x3.argtypes = [fp, fp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.oc_upconv2x2.argtypes = [fp, fp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.oc_head.argtypes = [fp, fp, ctypes.c_float, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        _LIB = lib
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def forward_chain(weights, image):
    """image (H,W) -> logits (H,W) float32, bit-for-bit the product's f32 MFMA path."""
    lib = _chain_lib()
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    d = depth_of(weights)
    x = f32(image)[:, :, None]

    def conv(x, name, relu=1):
        w, b = f32(weights[name + "_w"]), f32(weights[name + "_b"])
        H, W, Cin = x.shape
        out = np.empty((H, W, w.shape[3]), dtype=np.float32)
        lib.oc_conv3x3(_p(x), _p(w), _p(b), _p(out), H, W, Cin, w.shape[3], relu)
        return out

    skips = []
    for i in range(d):
        x = conv(conv(x, f"enc{i}a"), f"enc{i}b")
        skips.append(x)
        x = np.ascontiguousarray(_pool(x))
    x = conv(conv(x, "bota"), "botb")
    for i in reversed(range(d)):
        w, b = f32(weights[f"up{i}_w"]), f32(weights[f"up{i}_b"])
        H, W, Cin = x.shape
        up = np.empty((2 * H, 2 * W, w.shape[3]), dtype=np.float32)
        lib.oc_upconv2x2(_p(x), _p(w), _p(b), _p(up), H, W, Cin, w.shape[3])
        x = np.ascontiguousarray(np.concatenate([skips[i], up], axis=2))
        x = conv(conv(x, f"dec{i}a"), f"dec{i}b")
    H, W, C = x.shape
    out = np.empty((H, W), dtype=np.float32)
    lib.oc_head(_p(x), _p(f32(weights["head_w"])), ctypes.c_float(float(weights["head_b"])), _p(out), H, W, C)
    return out
