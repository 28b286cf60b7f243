"""Import the parameters of a UNet stored as an ONNX file.

The reference runs `humerus/models/unetcrf_anp.onnx` through onnxruntime
(`src/shoulder/humerus/anatomic_neck.py:62-76`).  That blob is missing from the reference
tree, so its graph cannot be known here; this importer accepts the family of networks the
engine executes (`sh_load_unet`, include/shoulder_hip.h) and refuses everything else loudly,
naming the first node it cannot map:

  input float32[N,1,H,W]
  depth x { Conv3x3 [BatchNorm] Relu, Conv3x3 [BatchNorm] Relu, MaxPool2x2 }
  Conv3x3 [BN] Relu, Conv3x3 [BN] Relu                                    (bottleneck)
  depth x { ConvTranspose2x2/s2, Concat(skip, up) in either order, Conv3x3 [BN] Relu, Conv3x3 [BN] Relu }
  Conv1x1 -> one logit channel (the reference thresholds the output at 0, anatomic_neck.py:82)

BatchNormalization nodes (inference form) are folded into the convolution before them.
No onnx / onnxruntime package is needed: the file is read with a minimal protobuf wire
reader (field numbers of onnx.proto3: ModelProto.graph=7; GraphProto.node=1,
initializer=5, input=11, output=12; NodeProto input=1 output=2 name=3 op_type=4
attribute=5; AttributeProto name=1 f=2 i=3 s=4 floats=7 ints=8; TensorProto dims=1
data_type=2 float_data=4 name=8 raw_data=9 double_data=10).
"""
import struct

import numpy as np


class UnsupportedOnnxModel(ValueError):
    """The graph is not a network `sh_load_unet` can execute."""


# ---- protobuf wire format ------------------------------------------------------------------------
def _varint(b, i):
    r = s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        if not c & 0x80:
            return r, i
        s += 7


def _fields(b):
    i, n = 0, len(b)
    while i < n:
        tag, i = _varint(b, i)
        fn, wt = tag >> 3, tag & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 1:
            v, i = b[i:i + 8], i + 8
        elif wt == 2:
            ln, i = _varint(b, i)
            v, i = b[i:i + ln], i + ln
        elif wt == 5:
            v, i = b[i:i + 4], i + 4
        else:
            raise UnsupportedOnnxModel(f"protobuf wire type {wt} (field {fn}) is not expected in an ONNX file")
        if i > n:
            raise UnsupportedOnnxModel("truncated ONNX file")
        yield fn, wt, v


def _signed(x):
    return x - (1 << 64) if x >= (1 << 63) else x


def _ints(wt, v):
    if wt == 0:
        return [_signed(v)]
    out, j = [], 0
    while j < len(v):
        x, j = _varint(v, j)
        out.append(_signed(x))
    return out


_DTYPES = {1: np.float32, 10: np.float16, 11: np.float64, 7: np.int64, 6: np.int32}


def _tensor(b):
    dims, dt, name, raw, fdata, ddata = [], 1, "", None, [], []
    for fn, wt, v in _fields(b):
        if fn == 1:
            dims += _ints(wt, v)
        elif fn == 2:
            dt = v
        elif fn == 8:
            name = bytes(v).decode()
        elif fn == 9:
            raw = bytes(v)
        elif fn == 4:
            fdata += list(struct.unpack(f"<{len(v) // 4}f", bytes(v)))
        elif fn == 10:
            ddata += list(struct.unpack(f"<{len(v) // 8}d", bytes(v)))
        elif fn in (13, 14) and (fn == 13 or v == 1):
            raise UnsupportedOnnxModel(f"initializer {name!r} keeps its data in an external file")
    if dt not in _DTYPES:
        raise UnsupportedOnnxModel(f"initializer {name!r}: data type {dt} is not supported")
    if raw is not None:
        a = np.frombuffer(raw, dtype=np.dtype(_DTYPES[dt]).newbyteorder("<"))
    elif dt == 11:
        a = np.asarray(ddata, dtype=np.float64)
    else:
        a = np.asarray(fdata, dtype=np.float32)
    n = int(np.prod(dims)) if dims else 1
    if a.size != n:
        raise UnsupportedOnnxModel(f"initializer {name!r}: {a.size} values for shape {dims}")
    return name, a.astype(np.float64 if dt == 11 else np.float32 if dt in (1, 10) else a.dtype).reshape(dims)


def _attribute(b):
    name, val = "", None
    for fn, wt, v in _fields(b):
        if fn == 1:
            name = bytes(v).decode()
        elif fn == 2:
            val = struct.unpack("<f", bytes(v))[0]
        elif fn == 3:
            val = _signed(v)
        elif fn == 4:
            val = bytes(v).decode(errors="replace")
        elif fn == 7:
            val = (val or []) + (list(struct.unpack(f"<{len(v) // 4}f", bytes(v))) if wt == 2 else [struct.unpack("<f", bytes(v))[0]])
        elif fn == 8:
            val = (val or []) + _ints(wt, v)
    return name, val


def _node(b):
    nd = {"input": [], "output": [], "name": "", "op": "", "attr": {}}
    for fn, wt, v in _fields(b):
        if fn == 1:
            nd["input"].append(bytes(v).decode())
        elif fn == 2:
            nd["output"].append(bytes(v).decode())
        elif fn == 3:
            nd["name"] = bytes(v).decode()
        elif fn == 4:
            nd["op"] = bytes(v).decode()
        elif fn == 5:
            k, a = _attribute(v)
            nd["attr"][k] = a
    return nd


def read_graph(path_or_bytes):
    """-> (nodes in file order, {initializer name: ndarray}, graph input names, graph output names)."""
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    graph = None
    for fn, wt, v in _fields(memoryview(data)):
        if fn == 7 and wt == 2:
            graph = v
    if graph is None:
        raise UnsupportedOnnxModel("no GraphProto in the file (not an ONNX model?)")
    nodes, inits, gin, gout = [], {}, [], []
    for fn, wt, v in _fields(graph):
        if fn == 1:
            nodes.append(_node(v))
        elif fn == 5:
            k, a = _tensor(v)
            inits[k] = a
        elif fn in (11, 12):
            nm = next((bytes(x).decode() for f2, _, x in _fields(v) if f2 == 1), "")
            (gin if fn == 11 else gout).append(nm)
    gin = [n for n in gin if n not in inits]          # old exporters list the initializers as inputs too
    return nodes, inits, gin, gout


# ---- graph -> sh_load_unet parameters ------------------------------------------------------------
def _where(nd):
    return f"node {nd['name'] or nd['output'][:1]} ({nd['op']})"


def _conv_params(nd, inits, transposed=False):
    a = nd["attr"]
    if len(nd["input"]) < 2 or nd["input"][1] not in inits:
        raise UnsupportedOnnxModel(f"{_where(nd)}: weights are not a constant initializer")
    W = np.asarray(inits[nd["input"][1]], dtype=np.float64)
    if W.ndim != 4:
        raise UnsupportedOnnxModel(f"{_where(nd)}: weight rank {W.ndim}, expected 4")
    k = W.shape[2]
    if W.shape[2] != W.shape[3] or a.get("group", 1) != 1 or any(d != 1 for d in a.get("dilations", [1, 1])):
        raise UnsupportedOnnxModel(f"{_where(nd)}: only square, dense, undilated kernels are supported")
    if a.get("auto_pad", "NOTSET") not in ("NOTSET", ""):
        raise UnsupportedOnnxModel(f"{_where(nd)}: auto_pad={a['auto_pad']} is not supported (explicit pads only)")
    pads, strides = a.get("pads", [0, 0, 0, 0]), a.get("strides", [1, 1])
    if transposed:
        if k != 2 or strides != [2, 2] or any(pads) or any(a.get("output_padding", [0, 0])):
            raise UnsupportedOnnxModel(f"{_where(nd)}: only ConvTranspose 2x2, stride 2, no padding is supported")
        cin, cout = W.shape[0], W.shape[1]
        Wk = W.transpose(2, 3, 0, 1)                  # [cin, cout, ky, kx] -> [ky][kx][cin][cout]
    else:
        if k not in (1, 3) or strides != [1, 1] or pads != [k // 2] * 4:
            raise UnsupportedOnnxModel(f"{_where(nd)}: only Conv 3x3/pad 1 and 1x1/pad 0 at stride 1 are supported (k={k}, pads={pads}, strides={strides})")
        cout, cin = W.shape[0], W.shape[1]
        Wk = W.transpose(2, 3, 1, 0)                  # [cout, cin, ky, kx] -> [ky][kx][cin][cout]
    if len(nd["input"]) > 2 and nd["input"][2]:
        if nd["input"][2] not in inits:
            raise UnsupportedOnnxModel(f"{_where(nd)}: bias is not a constant initializer")
        b = np.asarray(inits[nd["input"][2]], dtype=np.float64).reshape(-1)
    else:
        b = np.zeros(cout)
    return {"k": k, "cin": cin, "cout": cout, "w": Wk, "b": b, "transposed": transposed, "relu": False, "src": nd["input"][0]}


def unet_from_onnx(path_or_bytes):
    """-> (weights, base_channels, depth): `weights` has the keys/shapes of `unet_spec.make_teacher_weights`
    (conv weights [ky][kx][cin][cout] float32) and goes straight into `Engine.load_unet`."""
    nodes, inits, gin, gout = read_graph(path_or_bytes)
    if len(gin) != 1 or len(gout) != 1:
        raise UnsupportedOnnxModel(f"expected one graph input and one output, found {gin} -> {gout}")
    # every tensor name -> the layer (or tag) that produced it
    prod = {gin[0]: {"kind": "input"}}
    convs = []                                        # conv / up-conv layers in execution order
    for nd in nodes:
        op, x = nd["op"], nd["input"][0] if nd["input"] else None
        if op == "Constant":
            raise UnsupportedOnnxModel(f"{_where(nd)}: constants must be initializers")
        if x not in prod and op != "Concat":
            raise UnsupportedOnnxModel(f"{_where(nd)}: input {x!r} is not produced by a supported node")
        if op in ("Conv", "ConvTranspose"):
            L = _conv_params(nd, inits, transposed=(op == "ConvTranspose"))
            L["kind"] = "conv"
            convs.append(L)
            prod[nd["output"][0]] = L
        elif op == "BatchNormalization":
            L = prod[x]
            if L.get("kind") != "conv" or L["relu"] or L.get("consumed"):
                raise UnsupportedOnnxModel(f"{_where(nd)}: BatchNormalization must directly follow a convolution")
            try:
                g, beta, mu, var = (np.asarray(inits[n], dtype=np.float64).reshape(-1) for n in nd["input"][1:5])
            except KeyError as e:
                raise UnsupportedOnnxModel(f"{_where(nd)}: parameter {e} is not a constant initializer") from None
            s = g / np.sqrt(var + float(nd["attr"].get("epsilon", 1e-5)))
            L["w"] = L["w"] * s                       # last axis is cout
            L["b"] = (L["b"] - mu) * s + beta
            prod[nd["output"][0]] = L
        elif op == "Relu":
            L = prod[x]
            if L.get("kind") != "conv" or L["relu"]:
                raise UnsupportedOnnxModel(f"{_where(nd)}: Relu must follow a convolution (+BatchNormalization)")
            L["relu"] = True
            prod[nd["output"][0]] = L
        elif op == "MaxPool":
            a = nd["attr"]
            if a.get("kernel_shape") != [2, 2] or a.get("strides", [1, 1]) != [2, 2] or any(a.get("pads", [0] * 4)) or a.get("ceil_mode", 0):
                raise UnsupportedOnnxModel(f"{_where(nd)}: only MaxPool 2x2, stride 2 is supported")
            prod[nd["output"][0]] = {"kind": "pool", "of": prod[x]}
        elif op == "Concat":
            if nd["attr"].get("axis") != 1 or len(nd["input"]) != 2 or any(i not in prod for i in nd["input"]):
                raise UnsupportedOnnxModel(f"{_where(nd)}: only a channel concat of two produced tensors is supported")
            prod[nd["output"][0]] = {"kind": "cat", "parts": [prod[i] for i in nd["input"]]}
        elif op in ("Identity", "Dropout"):
            prod[nd["output"][0]] = prod[x]
        elif op == "Sigmoid":
            # Only as the tail of the network: the reference turns the model's output into a mask by a threshold
            # (anatomic_neck.py:79-83: `mask > 0` for its logit-output models, `mask > 0.5` for the sigmoid-output "b loss"
            # variant it keeps as a comment) and sigmoid(x) > 0.5 <=> x > 0, so the engine -- which thresholds the last
            # convolution's output at 0 -- computes the mask of a sigmoid-tailed export by dropping the tail.
            L = prod[x]
            if L.get("kind") != "conv" or L["relu"] or nd["output"][0] != gout[0]:
                raise UnsupportedOnnxModel(f"{_where(nd)}: Sigmoid is only accepted as the tail behind the last convolution")
            import warnings
            warnings.warn(f"{_where(nd)}: Sigmoid tail dropped -- the mask is sigmoid(x) > 0.5, i.e. logit > 0 "
                          "(the reference's `mask > 0.5` variant, anatomic_neck.py:81-82)", stacklevel=2)
            prod[nd["output"][0]] = L
        else:
            raise UnsupportedOnnxModel(f"{_where(nd)}: operator {op} is not part of the UNet family the engine executes "
                                       "(Conv, ConvTranspose, BatchNormalization, Relu, MaxPool, Concat; Sigmoid / Identity as the tail)")
    if gout[0] not in prod or prod[gout[0]].get("kind") != "conv":
        raise UnsupportedOnnxModel("the graph output is not a convolution (the reference thresholds raw logits at 0)")

    n3 = [L for L in convs if not L["transposed"] and L["k"] == 3]
    ups = [L for L in convs if L["transposed"]]
    heads = [L for L in convs if not L["transposed"] and L["k"] == 1]
    depth = len(ups)
    if depth < 1 or len(n3) != 4 * depth + 2 or len(heads) != 1 or heads[0] is not convs[-1] or prod[gout[0]] is not heads[0]:
        raise UnsupportedOnnxModel(f"not a double-conv UNet: {len(n3)} 3x3 convs, {len(ups)} up-convs, {len(heads)} 1x1 convs")
    base = n3[0]["cout"]
    ch = [base << i for i in range(depth + 1)]

    def src_of(L):
        return prod[L["src"]]

    def need(cond, what):
        if not cond:
            raise UnsupportedOnnxModel(f"not the expected UNet topology: {what}")

    w = {}

    def put(name, L, cin, cout, relu=True, perm=None):
        need(L["cin"] == cin and L["cout"] == cout, f"{name}: {L['cin']}->{L['cout']} channels, expected {cin}->{cout}")
        need(L["relu"] == relu, f"{name}: {'missing' if relu else 'unexpected'} Relu")
        W = L["w"] if perm is None else L["w"][:, :, perm, :]
        w[name + "_w"] = np.ascontiguousarray(W, dtype=np.float32)
        w[name + "_b"] = np.ascontiguousarray(L["b"], dtype=np.float32)

    it = iter(convs)
    prev, skips, cin = prod[gin[0]], [], 1
    for i in range(depth):
        a, b = next(it), next(it)
        need(src_of(a) is prev and src_of(b) is a, f"enc{i}: convolutions are not chained")
        put(f"enc{i}a", a, cin, ch[i])
        put(f"enc{i}b", b, ch[i], ch[i])
        skips.append(b)
        cin = ch[i]
        prev = next((p for p in prod.values() if p.get("kind") == "pool" and p["of"] is b), None)
        need(prev is not None, f"enc{i}: no MaxPool after the second convolution")
    a, b = next(it), next(it)
    need(src_of(a) is prev and src_of(b) is a, "bottleneck: convolutions are not chained")
    put("bota", a, ch[depth - 1], ch[depth])
    put("botb", b, ch[depth], ch[depth])
    prev = b
    for i in reversed(range(depth)):
        u, a, b = next(it), next(it), next(it)
        need(u["transposed"] and src_of(u) is prev, f"up{i}: expected a ConvTranspose of the level below")
        put(f"up{i}", u, ch[i + 1], ch[i], relu=False)
        cat = src_of(a)
        need(cat.get("kind") == "cat" and {id(p) for p in cat["parts"]} == {id(skips[i]), id(u)}, f"dec{i}a: input is not Concat(enc{i}b, up{i})")
        # the engine concatenates [skip, up]; a graph that concatenates [up, skip] gets its input channels swapped
        perm = None if cat["parts"][0] is skips[i] else np.r_[ch[i]:2 * ch[i], 0:ch[i]]
        put(f"dec{i}a", a, 2 * ch[i], ch[i], perm=perm)
        need(src_of(b) is a, f"dec{i}b: convolutions are not chained")
        put(f"dec{i}b", b, ch[i], ch[i])
        prev = b
    h = next(it)
    need(src_of(h) is prev and h["cout"] == 1 and h["cin"] == ch[0], "head: expected Conv1x1 to one channel on dec0b")
    need(not h["relu"], "head: unexpected Relu")
    w["head_w"] = np.ascontiguousarray(h["w"].reshape(ch[0]), dtype=np.float32)
    w["head_b"] = np.float32(h["b"][0])
    return w, base, depth
