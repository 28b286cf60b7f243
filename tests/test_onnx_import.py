"""UNet parameter import from an ONNX file (reference anatomic_neck.py:62-76 runs the file through
onnxruntime).  The layout conventions are pinned against torch's conv2d / conv_transpose2d / max_pool2d, whose weight
layouts and semantics are those of the ONNX operators; the importer's output is then run by the oracle's UNet."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import unet as ounet
from shoulder_amd import unet_spec
from shoulder_amd.onnx_import import UnsupportedOnnxModel, read_graph, unet_from_onnx
import _onnx_writer as ow

BASE, DEPTH = 4, 2


def onnx_layout(w, depth, swap_cat=False):
    """engine-layout dict -> {layer: (W, b)} in ONNX layouts; swap_cat: the graph will concatenate [up, skip]."""
    out = {}
    for k in [k[:-2] for k in w if k.endswith("_w") and k != "head_w"]:
        W = np.asarray(w[k + "_w"], dtype=np.float32)
        if k.startswith("up"):
            out[k] = (W.transpose(2, 3, 0, 1), w[k + "_b"])
        else:
            if swap_cat and k.startswith("dec") and k.endswith("a"):
                c = W.shape[2] // 2
                W = np.concatenate([W[:, :, c:], W[:, :, :c]], axis=2)
            out[k] = (W.transpose(3, 2, 0, 1), w[k + "_b"])
    out["head"] = (np.asarray(w["head_w"], np.float32).reshape(1, -1, 1, 1), np.asarray([w["head_b"]], np.float32))
    return out


def torch_forward(ow_, depth, x, cat_skip_first=True, bn=()):
    """The graph `_onnx_writer.unet_model` writes, executed with torch ops on the ONNX-layout tensors."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).double()

    def conv(x, name, relu=True):
        W, b = ow_[name]
        y = F.conv2d(x, t(W), None if b is None else t(b), padding=W.shape[2] // 2)
        if name in bn:
            g, beta, mu, var, eps = bn[name]
            y = F.batch_norm(y, t(mu), t(var), t(g), t(beta), False, 0.0, eps)
        return F.relu(y) if relu else y

    x = torch.from_numpy(x).double()[None, None]
    skips = []
    for i in range(depth):
        x = conv(conv(x, f"enc{i}a"), f"enc{i}b")
        skips.append(x)
        x = F.max_pool2d(x, 2)
    x = conv(conv(x, "bota"), "botb")
    for i in reversed(range(depth)):
        W, b = ow_[f"up{i}"]
        u = F.conv_transpose2d(x, t(W), t(b), stride=2)
        x = torch.cat([skips[i], u] if cat_skip_first else [u, skips[i]], dim=1)
        x = conv(conv(x, f"dec{i}a"), f"dec{i}b")
    return conv(x, "head", relu=False)[0, 0].numpy()


@pytest.fixture(scope="module")
def weights():
    return unet_spec.make_teacher_weights(seed=7, base=BASE, depth=DEPTH)


@pytest.mark.parametrize("raw,as_inputs", [(True, False), (False, True)])
def test_round_trip_is_exact(weights, raw, as_inputs):
    blob = ow.unet_model(onnx_layout(weights, DEPTH), DEPTH, raw=raw, inits_as_inputs=as_inputs)
    w, base, depth = unet_from_onnx(blob)
    assert (base, depth) == (BASE, DEPTH)
    assert set(w) == set(weights)
    for k in weights:
        assert np.array_equal(np.asarray(w[k]), np.asarray(weights[k], dtype=np.float32)), k
        assert np.asarray(w[k]).shape == np.asarray(weights[k]).shape, k


@pytest.mark.parametrize("skip_first", [True, False])
def test_imported_network_computes_what_the_graph_computes(weights, skip_first):
    rng = np.random.default_rng(3)
    ow_ = onnx_layout(weights, DEPTH, swap_cat=False)
    # a graph with batch-norm after two convolutions (one of them bias-free) and either concat order
    bn = {}
    for name in ("enc1a", "dec0a"):
        c = ow_[name][0].shape[0]
        bn[name] = (rng.uniform(0.5, 1.5, c), rng.normal(0, 0.1, c), rng.normal(0, 0.1, c), rng.uniform(0.5, 2.0, c), 1e-3)
    ow_["enc1a"] = (ow_["enc1a"][0], None)
    blob = ow.unet_model(ow_, DEPTH, cat_skip_first=skip_first, bn=bn)
    w, base, depth = unet_from_onnx(blob)
    x = rng.uniform(0, 1, (16, 24)).astype(np.float32)
    want = torch_forward(ow_, DEPTH, x, cat_skip_first=skip_first, bn=bn)
    got = ounet.forward_f64(w, x)
    assert np.abs(np.asarray(got, dtype=np.float64) - want).max() < 2e-5 * max(1.0, np.abs(want).max())


def test_rejections_name_the_node(weights):
    ow_ = onnx_layout(weights, DEPTH)
    with pytest.raises(UnsupportedOnnxModel, match="Softmax"):
        unet_from_onnx(ow.unet_model(ow_, DEPTH, tail="Softmax"))
    bad = dict(ow_)
    W, b = bad["enc0b"]
    bad["enc0b"] = (np.zeros((W.shape[0], W.shape[1], 5, 5), np.float32), b)
    with pytest.raises(UnsupportedOnnxModel, match="enc0b"):
        unet_from_onnx(ow.unet_model(bad, DEPTH))
    bad = dict(ow_)
    W, b = bad["dec1b"]
    bad["dec1b"] = (W[:-1], b[:-1])
    with pytest.raises(UnsupportedOnnxModel, match="channels"):
        unet_from_onnx(ow.unet_model(bad, DEPTH))
    with pytest.raises(UnsupportedOnnxModel):
        unet_from_onnx(b"\x08\x08")
    with pytest.raises(UnsupportedOnnxModel):
        unet_from_onnx(ow.unet_model(ow_, DEPTH)[:-40])


def test_monotone_tails_are_dropped(weights):
    """A Sigmoid (or Identity) behind the head is the reference's thresholded-output variant (anatomic_neck.py:79-83):
    sigmoid(x) > 0.5 <=> x > 0, so the same parameters come out, with a warning that names the convention."""
    ow_ = onnx_layout(weights, DEPTH)
    plain, base, depth = unet_from_onnx(ow.unet_model(ow_, DEPTH))
    with pytest.warns(UserWarning, match="Sigmoid tail dropped"):
        sig, b2, d2 = unet_from_onnx(ow.unet_model(ow_, DEPTH, tail="Sigmoid"))
    ident, b3, d3 = unet_from_onnx(ow.unet_model(ow_, DEPTH, tail="Identity"))
    assert (base, depth) == (b2, d2) == (b3, d3)
    for k in plain:
        assert np.array_equal(plain[k], sig[k]) and np.array_equal(plain[k], ident[k]), k


def test_reader_lists_nodes_and_initializers(weights):
    nodes, inits, gin, gout = read_graph(ow.unet_model(onnx_layout(weights, DEPTH), DEPTH))
    assert gin == ["input"] and len(gout) == 1
    assert [n["op"] for n in nodes].count("Conv") == 4 * DEPTH + 3
    assert inits["up1.weight"].shape == (4 * BASE, 2 * BASE, 2, 2)


@pytest.mark.gpu
def test_engine_runs_the_imported_network():
    from shoulder_amd import _lib
    from shoulder_amd.engine import Engine
    rng = np.random.default_rng(5)
    weights = unet_spec.make_teacher_weights(seed=11, base=32, depth=2)      # the engine wants base % 32 == 0
    ow_ = onnx_layout(weights, 2)
    bn = {"bota": (rng.uniform(0.5, 1.5, 128), rng.normal(0, 0.1, 128), rng.normal(0, 0.1, 128), rng.uniform(0.5, 2.0, 128), 1e-5)}
    blob = ow.unet_model(ow_, 2, cat_skip_first=False, bn=bn)
    eng = Engine(0)
    assert eng.load_unet_onnx(blob) == (32, 2)
    eng.set_params(unet_dtype=_lib.UNET_F32)
    x = rng.uniform(0, 1, (2, 64, 128)).astype(np.float32)      # multiples of 16 << depth
    got = eng.unet_infer(x)
    for i in range(2):
        want = torch_forward(ow_, 2, x[i], cat_skip_first=False, bn=bn)
        assert np.abs(got[i] - want).max() < 1e-3 * max(1.0, np.abs(want).max())      # f32 MFMA (split-bf16 products) vs float64
