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
            cache[name] = OracleHumerus.from_stl(os.path.join(BONES, name + ".stl"), rfc_tables, unet_weights)
        return cache[name]
    return get


@pytest.fixture(scope="session")
def engine():
    from shoulder_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()
