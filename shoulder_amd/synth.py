"""Synthetic humerus batches: mesh i = S_i R_i (v - c) + c + t_i of a
template mesh; R_i uniform on SO(3) from a seeded unit quaternion, t_i ~ U(-500, 500)^3 mm,
S_i ~ U(0.85, 1.15), numpy.random.default_rng(1234).  No shear, no mirror."""
import numpy as np


def similarity_transforms(n, template_verts, seed=1234, start=0):
    """-> (n,4,4) float64 for meshes start..start+n-1 of the seeded sequence."""
    # three child streams of the seed (rotation, translation, scale): mesh i is the same whatever the
    # batch size or shard, so contiguous per-rank shards tile the global sequence exactly
    sq, st, ss = np.random.SeedSequence(seed).spawn(3)
    total = start + n
    q = np.random.default_rng(sq).standard_normal((total, 4))
    t = np.random.default_rng(st).uniform(-500.0, 500.0, (total, 3))
    s = np.random.default_rng(ss).uniform(0.85, 1.15, total)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    c = np.asarray(template_verts, dtype=np.float64).mean(axis=0)
    T = np.zeros((total, 4, 4))
    w, x, y, z = q.T
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                  np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                  np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], 1)
    A = s[:, None, None] * R
    T[:, :3, :3] = A
    T[:, :3, 3] = c + t - A @ c
    T[:, 3, 3] = 1.0
    return T[start:]


def apply_similarity(T, verts):
    """float64 arithmetic, float32 storage (like an STL)."""
    v = np.asarray(verts, dtype=np.float64)
    return ((v @ T[:3, :3].T) + T[:3, 3]).astype(np.float32)
