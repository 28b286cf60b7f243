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
