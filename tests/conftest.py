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
    import oracle as o
    o.lib()
    return o


PKG = os.path.join(ROOT, "tokamak-zk-evm_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)
TOOLS = os.path.join(ROOT, "tools")                # tools/synth_circuit.py: synthetic circuits in the reference's file formats
if TOOLS not in sys.path:
    sys.path.insert(0, TOOLS)


@pytest.fixture(scope="session")
def tkmk():
    """The product: ctypes binding over tokamak-zk-evm_amd/libtkmk_hip.so (C ABI in include/tkmk.h)."""
    import tkmk as t
    if not os.path.exists(t.LIB_PATH):      # normally built beforehand by __graft_entry__.build(); same recipe if it was not
        import shutil
        if shutil.which("hipcc") is None:
            pytest.fail("libtkmk_hip.so is not built and hipcc is not available")
        import __graft_entry__
        __graft_entry__.build()
    t.lib()
    return t


@pytest.fixture(scope="session")
def gpu(tkmk):
    if tkmk.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible (the product has no CPU fallback)")
    tkmk.set_device(0)
    return tkmk
