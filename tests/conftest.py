import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orb():
    """The product package (directory name has a hyphen, so it is imported by name)."""
    return importlib.import_module("orb-slam2_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("orb-slam2_amd.synth")


@pytest.fixture(scope="session")
def orbref():
    """ctypes binding of the CPU oracle (test infrastructure)."""
    import orbref as m
    m.lib()
    return m
