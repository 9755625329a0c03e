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
    # The GPU tests are about the library on the runtime it is linked against.  A process that imported torch first has
    # torch's bundled librccl / libamdhip64 underneath librtx_hip.so instead: say so rather than test something else.
    info = rt.comm_info()
    if info.count("/opt/rocm") != 2:
        pytest.fail("librtx_hip.so is not running on /opt/rocm's RCCL + HIP runtime (" + info + "): run the GPU tests in a "
                    "process of their own (pytest -m gpu), without torch imported")
    ctx = rt.Context(0)
    yield ctx
    ctx.close()
