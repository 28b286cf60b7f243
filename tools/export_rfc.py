"""Decode the reference's bicipital-groove random forest into flat tables.

Run ONCE in the build container (the reference tree is not present on the GPU box):
    python tools/export_rfc.py /root/reference/src/shoulder/humerus/models/rfc_bg3.onnx
Writes shoulder_amd/models/rfc_bg3.npz (model PARAMETERS -- data, not code):
feat int32[N], thr float32[N], true_idx/false_idx int32[N] (global node index, -1 at
leaves), leaf_weight float32[N] (P(class 1)/n_trees at leaves), roots int32[n_trees].
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.rfc import decode_onnx_forest  # noqa: E402

if __name__ == "__main__":
    t = decode_onnx_forest(sys.argv[1])
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shoulder_amd", "models", "rfc_bg3.npz")
    np.savez_compressed(out, feat=t["feat"], thr=t["thr"], true_idx=t["true_idx"], false_idx=t["false_idx"],
                        leaf_weight=t["leaf_weight"], roots=t["roots"])
    print(out, {k: (v.shape, v.dtype) for k, v in t.items()})
    print("branch", int((~t["is_leaf"]).sum()), "leaves", int(t["is_leaf"].sum()), "trees", len(t["roots"]),
          "feat counts", np.bincount(t["feat"][~t["is_leaf"]], minlength=9))
