import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the package directory name contains '-': load it with importlib and expose a short alias
rtow = importlib.import_module("raytracing-in-one-weekend_amd")
sys.modules.setdefault("rtow_amd", rtow)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # no test of this suite runs for minutes: one that does (a kernel that never ends, a library load on a sick box) fails after 10 minutes with
    # every thread's stack instead of holding the GPU box until the caller's own limit (pytest-timeout, where installed)
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600, method="thread"))


def _has_gpu():
    try:
        ctx = rtow.Context(0)
        ctx.close()
        return True
    except Exception:
        return False


@pytest.fixture(scope="session")
def rt():
    return rtow


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.load("strict")
    return binding


@pytest.fixture(scope="session")
def gpu_context():
    """A context on cuda:0.  GPU tests FAIL (not skip) when the HIP library or device is missing."""
    ctx = rtow.Context(0)
    yield ctx
    ctx.close()
