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
    """select a non-default kernel / schedule through the test-only entry point of include/sda_hip_debug.h.  The RELEASE library
    (libsda_hip.so) has no knob table: the first call makes libsda_hip_test.so - the same objects, sda_capi.cpp rebuilt with
    -DSDA_TEST_HOOKS - the active library of this process (sda_amd.capi.use_test_hooks); the fixture below resets the knobs and
    goes back to the release library after every test, so every test that does NOT touch a knob runs on the shipped binary."""
    from sda_amd import capi
    if isinstance(value, str):
        value = VARINT_PATHS[value] if name == "SDA_VARINT_PATH" else int(value)
    capi.use_test_hooks()
    capi.check(capi.load().sda_debug_set_knob(name.encode(), value))


def use_test_hooks():
    """tests that need the helpers of the test library without a knob (streams, memory figures, the selection table)"""
    from sda_amd import capi
    return capi.use_test_hooks()


@pytest.fixture(autouse=True)
def _reset_knobs():
    yield
    if "sda_amd.capi" in sys.modules:
        sys.modules["sda_amd.capi"].use_release()          # resets the knobs of the test library if it was active
