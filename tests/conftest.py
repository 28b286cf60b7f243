import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
BONES = os.path.join(GOLDEN, "bones")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rfc_tables():
    from oracle import rfc
    return rfc.load_tables(os.path.join(ROOT, "shoulder_amd", "models", "rfc_bg3.npz"))


def _teacher_weights():
    from shoulder_amd import unet_spec
    return unet_spec.make_teacher_weights()


def _ensure_chain_lib():
    import subprocess
    so = os.path.join(ROOT, "oracle", "_build", "libunet_chain.so")
    src = os.path.join(ROOT, "oracle", "unet_chain.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


import contextlib


@contextlib.contextmanager
def engine_with_env(**env):
    """An engine of its own, created while the given environment switches are set (the library reads its switches ONCE, when a context is
    created -- include/shoulder_hip.h): the A/B arm of a test.  Forest and teacher network loaded."""
    from shoulder_amd.engine import Engine
    from shoulder_amd import unet_spec
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        e = Engine(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        e.load_rfc()
        e.load_unet(_teacher_weights(), unet_spec.BASE, unet_spec.DEPTH)
        yield e
    finally:
        e.close()


@pytest.fixture(scope="session")
def unet_weights():
    return _teacher_weights()


@pytest.fixture(scope="session")
def oracle_bones(rfc_tables, unet_weights):
    """name -> OracleHumerus (lazy, memoised per session)."""
    from oracle.humerus import OracleHumerus
    cache = {}

    def get(name):
        if name not in cache:
            _ensure_chain_lib()
            cache[name] = OracleHumerus.from_stl(os.path.join(BONES, name + ".stl"), rfc_tables, unet_weights, unet_eval="chain")
        return cache[name]
    return get


@pytest.fixture(scope="session")
def engine():
    from shoulder_amd.engine import Engine
    e = Engine(0)
    from shoulder_amd import unet_spec
    e.load_rfc()
    e.load_unet(_teacher_weights(), unet_spec.BASE, unet_spec.DEPTH)
    yield e
    e.close()


def convex_surface(n, seed=0):
    """A strictly convex closed surface with n vertices, ALL of them on its convex hull: points spread over an egg-shaped
    ellipsoid (semi-axes 18 x 24 x 120 mm, tapered along its length so that the two ends differ; spiral points + jitter, so no two
    candidate boxes tie), rotated and shifted like a CT scan, rounded to float32 like an STL, triangulated by its own hull
    (outward windings).  -> (verts float32 [n, 3], faces int32 [2 n - 4, 3])"""
    import numpy as np
    import scipy.spatial
    rng = np.random.default_rng(seed)
    i = np.arange(n) + 0.5
    z = 1.0 - 2.0 * i / n
    phi = i * np.pi * (3.0 - np.sqrt(5.0)) + rng.uniform(-0.3, 0.3, n) / np.sqrt(n)
    z = np.clip(z + rng.uniform(-0.3, 0.3, n) / n, -1.0, 1.0)
    r = np.sqrt(1.0 - z * z)
    taper = 1.0 + 0.15 * z
    p = np.c_[18.0 * taper * r * np.cos(phi), 24.0 * taper * r * np.sin(phi), 120.0 * z]
    a, b, c = 0.4, -0.7, 1.1
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(c), -np.sin(c)], [0, np.sin(c), np.cos(c)]])
    v = (p @ (Rz @ Ry @ Rx).T + np.array([35.0, -80.0, 410.0])).astype(np.float32)
    h = scipy.spatial.ConvexHull(v.astype(np.float64), qhull_options="QbB Pp Qt")
    assert len(h.vertices) == n, f"{n - len(h.vertices)} of {n} points are not on the hull"
    tri = h.simplices.copy()
    ctr = v.astype(np.float64).mean(axis=0)
    p0, p1, p2 = (v[tri[:, k]].astype(np.float64) for k in range(3))
    inward = np.einsum("ij,ij->i", np.cross(p1 - p0, p2 - p0), p0 - ctr) < 0
    tri[inward] = tri[inward][:, [0, 2, 1]]
    return v, tri.astype(np.int32)


def lens_surface(nrim, ncap, seed=0, size=1.0):
    """A strictly convex lens (ellipsoid 60 x 40 x 8 mm x size) with `nrim` of its vertices ON its equator: seen along its short axis
    the silhouette is that ring -- more edges than an LDS tier of k_obb_candidates lists.  -> (verts float32, faces int32), as
    convex_surface."""
    import numpy as np
    import scipy.spatial
    rng = np.random.default_rng(seed)
    t = (np.arange(nrim) + rng.uniform(-0.2, 0.2, nrim)) * (2.0 * np.pi / nrim)
    rim = np.c_[np.cos(t), np.sin(t), np.zeros(nrim)]
    i = np.arange(ncap) + 0.5
    z = 1.0 - 2.0 * i / ncap
    z = z[np.abs(z) > 0.08]
    phi = np.arange(len(z)) * np.pi * (3.0 - np.sqrt(5.0)) + rng.uniform(-0.2, 0.2, len(z))
    r = np.sqrt(1.0 - z * z)
    cap = np.c_[r * np.cos(phi), r * np.sin(phi), z]
    p = np.concatenate([rim, cap]) * np.array([60.0, 40.0, 8.0]) * size
    a, b = 0.3, 0.9
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    v = (p @ (Rz @ Rx).T + np.array([-20.0, 55.0, 130.0])).astype(np.float32)
    h = scipy.spatial.ConvexHull(v.astype(np.float64), qhull_options="QbB Pp Qt")
    assert len(h.vertices) == len(v), f"{len(v) - len(h.vertices)} of {len(v)} points are not on the hull"
    tri = h.simplices.copy()
    ctr = v.astype(np.float64).mean(axis=0)
    p0, p1, p2 = (v[tri[:, k]].astype(np.float64) for k in range(3))
    inward = np.einsum("ij,ij->i", np.cross(p1 - p0, p2 - p0), p0 - ctr) < 0
    tri[inward] = tri[inward][:, [0, 2, 1]]
    return v, tri.astype(np.int32)
