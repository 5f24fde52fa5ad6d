import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle import pyoracle
    if not pyoracle.have_ref() and not os.path.isdir("/root/reference"):
        pytest.skip("oracle/_ref not built and /root/reference absent")
    return pyoracle.Reference()


@pytest.fixture(scope="session")
def manifest():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "manifest.json")) as f:
        return json.load(f)


def golden_path(name):
    return os.path.join(ROOT, "tests", "golden", name)
