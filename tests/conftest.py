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
    config.addinivalue_line("markers", "dense_lib: the plain dense GEMMs may go to hipBLASLt (csrc/wqaa_dense_lib.hip), the product default")


@pytest.fixture(autouse=True)
def own_dense_members_unless_asked(request, monkeypatch):
    """Plain dense GEMMs (W_dtype == A_dtype, M >= 16, no bias) go to hipBLASLt by default.  The parity tests of this suite
    are about the library's OWN kernels, so they pin the dense shapes to the own members (WQAA_DENSE_LIB=0, a plan-time
    switch); tests of the vendor-library path itself carry `@pytest.mark.dense_lib` (tests/test_dense_lib_gpu.py)."""
    if request.node.get_closest_marker("dense_lib") is None:
        monkeypatch.setenv("WQAA_DENSE_LIB", "0")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "packing_golden.npz"))
