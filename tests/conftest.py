import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    """The product binding (rust-raytracer_amd); builds nothing -- the .so must exist."""
    return importlib.import_module("rust-raytracer_amd")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle binding; (re)builds liboracle.so from oracle/ when a compiler is there."""
    import orc as o
    if not os.path.exists(o.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return o


@pytest.fixture(scope="session")
def gpu(rt):
    ctx = rt.Context(0)
    yield ctx
    ctx.close()
