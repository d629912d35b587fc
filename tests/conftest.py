import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def built():
    """libsda_hip.so + the C oracle, built in-tree (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def gpu(built):
    from sda_amd import capi
    lib = capi.load()
    if lib.sda_device_count() < 1:
        pytest.fail("-m gpu tests need a GPU and the native library; none visible "
                    "(there is no CPU fallback to run them on)")
    return lib


VARINT_PATHS = {"stream": 1, "scan": 2}


def set_knob(name, value=1):
    """select a non-default kernel / schedule through the library's test-only entry point (include/sda_hip_debug.h); the
    release library reads no environment variable.  Reset after every test by the fixture below."""
    from sda_amd import capi
    if isinstance(value, str):
        value = VARINT_PATHS[value] if name == "SDA_VARINT_PATH" else int(value)
    capi.check(capi.load().sda_debug_set_knob(name.encode(), value))


@pytest.fixture(autouse=True)
def _reset_knobs():
    yield
    if "sda_amd.capi" in sys.modules:
        lib = sys.modules["sda_amd.capi"]._lib if hasattr(sys.modules["sda_amd.capi"], "_lib") else None
        if lib is not None:
            lib.sda_debug_reset_knobs()
