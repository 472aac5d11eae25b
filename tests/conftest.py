import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu`)")
    config.addinivalue_line("markers", "selector_choice: tests/test_gemvx_gpu.py - the test checks what the selector picks on its own")
    config.addinivalue_line("markers", "dense_lib: opts into the vendor-library yardstick (WQAA_DENSE_LIB=1, csrc/wqaa_dense_lib.hip)")


@pytest.fixture(autouse=True)
def vendor_library_only_where_asked(request, monkeypatch):
    """Every operator runs this library's own kernels by default - the suite tests what ships.  hipBLASLt is an opt-in
    yardstick (WQAA_DENSE_LIB=1, a plan-time switch: plain dense pairs, and the second pass of the two-pass member); the tests
    of that path carry `@pytest.mark.dense_lib` (tests/test_dense_lib_gpu.py, tests/test_two_pass_gpu.py)."""
    if request.node.get_closest_marker("dense_lib") is not None:
        monkeypatch.setenv("WQAA_DENSE_LIB", "1")
    else:
        monkeypatch.delenv("WQAA_DENSE_LIB", raising=False)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "packing_golden.npz"))
