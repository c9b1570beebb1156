import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def tuning():
    """Pins which of the bit-identical kernel decompositions the C ABI dispatches to (mcn_set_tuning); calls
    accumulate within a test, the previous settings come back afterwards."""
    import ctypes
    from modelcrowdnav_amd import _hip
    prev, cur = _hip.get_tuning(), {}

    def set_(**kw):
        cur.update(kw)
        _hip.set_tuning(**cur)
    yield set_
    _hip.check(_hip.lib.mcn_set_tuning(ctypes.byref(prev)), "mcn_set_tuning")
