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
