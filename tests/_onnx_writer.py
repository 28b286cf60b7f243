"""Minimal ONNX (protobuf wire) writer for the importer tests: emits a UNet graph in the conventions of the usual
exporters (Conv weights [cout,cin,kh,kw], ConvTranspose weights [cin,cout,kh,kw], initializers as raw_data or
float_data).  Test infrastructure only."""
import struct

import numpy as np


def _vi(x):
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        c = x & 0x7F
        x >>= 7
        out.append(c | (0x80 if x else 0))
        if not x:
            return bytes(out)


def _ld(fn, payload):
    return _vi(fn << 3 | 2) + _vi(len(payload)) + bytes(payload)


def _v(fn, x):
    return _vi(fn << 3) + _vi(x)


def tensor(name, a, raw=True):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = b"".join(_v(1, d) for d in a.shape) + _v(2, 1)
    b += _ld(9, a.tobytes()) if raw else _ld(4, a.tobytes())      # packed float_data has the same bytes
    return b + _ld(8, name.encode())


def attr_ints(name, ints):
    return _ld(1, name.encode()) + _ld(8, b"".join(_vi(i) for i in ints)) + _v(20, 7)


def attr_int(name, i):
    return _ld(1, name.encode()) + _v(3, i) + _v(20, 2)


def attr_float(name, f):
    return _ld(1, name.encode()) + _vi(2 << 3 | 5) + struct.pack("<f", f) + _v(20, 1)


def node(op, inputs, outputs, attrs=(), name=""):
    b = b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs)
    b += _ld(3, (name or outputs[0]).encode()) + _ld(4, op.encode())
    return b + b"".join(_ld(5, a) for a in attrs)


def value_info(name):
    return _ld(1, name.encode())


def model(nodes, inits, inputs, outputs):
    g = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"g") + b"".join(_ld(5, t) for t in inits)
    g += b"".join(_ld(11, value_info(i)) for i in inputs) + b"".join(_ld(12, value_info(o)) for o in outputs)
    return _v(1, 8) + _ld(2, b"shoulder-tests") + _ld(7, g)


def unet_model(onnx_w, depth, cat_skip_first=True, bn=(), raw=True, inits_as_inputs=False, tail=None):
    """onnx_w: {layer: (W, b)} in ONNX layouts for enc{i}a/b, bota/b, up{i}, dec{i}a/b, head; `bn`: {layer: (g, beta, mu, var, eps)}
    inserted between that convolution and its Relu."""
    nodes, inits = [], []
    cnt = [0]

    def fresh(p):
        cnt[0] += 1
        return f"{p}_{cnt[0]}"

    def conv(x, name, k, relu=True, transposed=False):
        W, b = onnx_w[name]
        ins = [x, name + ".weight"]
        inits.append(tensor(name + ".weight", W, raw))
        if b is not None:
            inits.append(tensor(name + ".bias", b, raw))
            ins.append(name + ".bias")
        y = fresh(name)
        if transposed:
            attrs = [attr_ints("kernel_shape", [2, 2]), attr_ints("strides", [2, 2]), attr_ints("pads", [0, 0, 0, 0])]
        else:
            attrs = [attr_ints("kernel_shape", [k, k]), attr_ints("strides", [1, 1]), attr_ints("pads", [k // 2] * 4), attr_ints("dilations", [1, 1]), attr_int("group", 1)]
        nodes.append(node("ConvTranspose" if transposed else "Conv", ins, [y], attrs))
        if name in bn:
            g, beta, mu, var, eps = bn[name]
            for s, a in (("g", g), ("b", beta), ("m", mu), ("v", var)):
                inits.append(tensor(f"{name}.bn.{s}", a, raw))
            z = fresh(name + "_bn")
            nodes.append(node("BatchNormalization", [y] + [f"{name}.bn.{s}" for s in "gbmv"], [z], [attr_float("epsilon", eps)]))
            y = z
        if relu:
            z = fresh(name + "_relu")
            nodes.append(node("Relu", [y], [z]))
            y = z
        return y

    x, skips = "input", []
    for i in range(depth):
        x = conv(conv(x, f"enc{i}a", 3), f"enc{i}b", 3)
        skips.append(x)
        p = fresh("pool")
        nodes.append(node("MaxPool", [x], [p], [attr_ints("kernel_shape", [2, 2]), attr_ints("strides", [2, 2])]))
        x = p
    x = conv(conv(x, "bota", 3), "botb", 3)
    for i in reversed(range(depth)):
        u = conv(x, f"up{i}", 2, relu=False, transposed=True)
        c = fresh("cat")
        nodes.append(node("Concat", [skips[i], u] if cat_skip_first else [u, skips[i]], [c], [attr_int("axis", 1)]))
        x = conv(conv(c, f"dec{i}a", 3), f"dec{i}b", 3)
    x = conv(x, "head", 1, relu=False)
    if tail:
        y = fresh(tail)
        nodes.append(node(tail, [x], [y]))
        x = y
    ins = ["input"] + ([n for n in _names(inits)] if inits_as_inputs else [])
    return model(nodes, inits, ins, [x])


def _names(inits):
    from shoulder_amd.onnx_import import _tensor
    return [_tensor(memoryview(t))[0] for t in inits]
