"""K15: random-forest inference for the bicipital-groove classifier (oracle; test infrastructure).

Restates what onnxruntime 1.21.0 `TreeEnsembleClassifier` (absent from this image)
computes for `rfc_bg3.onnx` as called at reference
`src/shoulder/humerus/bicipital_groove.py:174-181`: input X double[N,9]; at a
`BRANCH_LEQ` node go to the true child iff x[feat] <= thr (thr = stored f32 widened);
score = sum of the reached leaves' class-0-id weights; probabilities = [1-s, s] f32.
`decode_onnx_forest` is a minimal protobuf wire reader (SURVEY App. C field numbers);
it is only run by tools/export_rfc.py to produce the table file shipped with the
product, and by tests to check that file against the model.
"""
import struct

import numpy as np


def _varint(b, i):
    r = 0
    s = 0
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
            v = b[i:i + 8]
            i += 8
        elif wt == 2:
            ln, i = _varint(b, i)
            v = b[i:i + ln]
            i += ln
        elif wt == 5:
            v = b[i:i + 4]
            i += 4
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield fn, wt, v


def _attr(b):
    name, out = None, {"floats": [], "ints": [], "strings": []}
    for fn, wt, v in _fields(b):
        if fn == 1:
            name = bytes(v).decode()
        elif fn == 7:
            if wt == 2:
                out["floats"].extend(struct.unpack(f"<{len(v) // 4}f", bytes(v)))
            else:
                out["floats"].append(struct.unpack("<f", bytes(v))[0])
        elif fn == 8:
            if wt == 2:
                j = 0
                while j < len(v):
                    x, j = _varint(v, j)
                    out["ints"].append(x - (1 << 64) if x >= (1 << 63) else x)
            else:
                out["ints"].append(v)
        elif fn == 9:
            out["strings"].append(bytes(v).decode())
    return name, out


def decode_onnx_forest(path):
    """-> dict of flat tables (global node indices) for the TreeEnsembleClassifier node."""
    data = memoryview(open(path, "rb").read())
    graph = next(v for fn, _, v in _fields(data) if fn == 7)
    attrs = None
    for fn, _, node in _fields(graph):
        if fn != 1:
            continue
        op, a = None, {}
        for nfn, _, nv in _fields(node):
            if nfn == 4:
                op = bytes(nv).decode()
            elif nfn == 5:
                k, val = _attr(nv)
                a[k] = val
        if op == "TreeEnsembleClassifier":
            attrs = a
    if attrs is None:
        raise ValueError("no TreeEnsembleClassifier node")
    tid = np.array(attrs["nodes_treeids"]["ints"], dtype=np.int64)
    nid = np.array(attrs["nodes_nodeids"]["ints"], dtype=np.int64)
    feat = np.array(attrs["nodes_featureids"]["ints"], dtype=np.int32)
    thr = np.array(attrs["nodes_values"]["floats"], dtype=np.float32)
    modes = attrs["nodes_modes"]["strings"]
    tnid = np.array(attrs["nodes_truenodeids"]["ints"], dtype=np.int64)
    fnid = np.array(attrs["nodes_falsenodeids"]["ints"], dtype=np.int64)
    assert set(modes) <= {"BRANCH_LEQ", "LEAF"}
    is_leaf = np.array([m == "LEAF" for m in modes])
    gidx = {(int(t), int(n)): i for i, (t, n) in enumerate(zip(tid, nid))}
    true_i = np.array([-1 if l else gidx[(int(t), int(n))] for l, t, n in zip(is_leaf, tid, tnid)], dtype=np.int32)
    false_i = np.array([-1 if l else gidx[(int(t), int(n))] for l, t, n in zip(is_leaf, tid, fnid)], dtype=np.int32)
    leafw = np.zeros(len(tid), dtype=np.float32)
    c_t = attrs["class_treeids"]["ints"]
    c_n = attrs["class_nodeids"]["ints"]
    c_id = attrs["class_ids"]["ints"]
    c_w = attrs["class_weights"]["floats"]
    assert all(c == 0 for c in c_id)
    for t, n, w in zip(c_t, c_n, c_w):
        leafw[gidx[(int(t), int(n))]] += np.float32(w)
    trees = np.unique(tid)
    roots = np.array([gidx[(int(t), 0)] for t in trees], dtype=np.int32)
    return dict(feat=feat, thr=thr, true_idx=true_i, false_idx=false_i, leaf_weight=leafw,
                is_leaf=is_leaf, roots=roots)


def load_tables(npz_path):
    z = np.load(npz_path)
    return {k: z[k] for k in z.files}


def predict_proba1(tables, X):
    """-> float32 P(class 1) per row (double accumulation in tree order, then f32)."""
    X = np.asarray(X, dtype=np.float64)
    feat, thr = tables["feat"], tables["thr"].astype(np.float64)
    ti, fi, lw = tables["true_idx"], tables["false_idx"], tables["leaf_weight"].astype(np.float64)
    n = len(X)
    s = np.zeros(n)
    rows = np.arange(n)
    for root in tables["roots"]:
        cur = np.full(n, root, dtype=np.int64)
        active = ti[cur] >= 0
        while active.any():
            c = cur[active]
            go_true = X[rows[active], feat[c]] <= thr[c]
            cur[active] = np.where(go_true, ti[c], fi[c])
            active = ti[cur] >= 0
        s += lw[cur]
    return s.astype(np.float32)
