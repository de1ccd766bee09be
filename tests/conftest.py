import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """host layer + oracle are plain g++ builds (seconds); the HIP engine is prebuilt in-tree by __graft_entry__.build()"""
    need = [os.path.join(ROOT, "skirt9_amd", "lib", "libskirthost.so"), os.path.join(ROOT, "oracle", "_build", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-s", "host", "oracle"], cwd=ROOT)
    yield


def ski(name):
    return os.path.join(ROOT, "tests", "ski", name)


def golden(name):
    return os.path.join(ROOT, "tests", "golden", name)


@pytest.fixture(autouse=True)
def _no_tuning_left_behind():
    """the engine's tuning switches (include/pmc_tuning.h) are process-wide: whatever a test sets is gone before the next one"""
    yield
    engine = sys.modules.get("skirt9_amd.engine")
    if engine is not None and engine._lib is not None:
        engine.clear_tuning()
