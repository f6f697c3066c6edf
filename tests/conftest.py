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
    """The CPU oracle (test infrastructure; oracle/tolfg_oracle.h explains its parity status)."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def tolfg():
    """The product library through its Python host layer; building it needs hipcc, not a GPU."""
    import tol_amd
    if not os.path.exists(tol_amd.lib_path()):
        from tol_amd import build
        build.build()
    tol_amd.lib()
    return tol_amd


@pytest.fixture(scope="session")
def measure(tolfg):
    """The measurement build of the library (tol_amd/lib/libtolfg_measure.so): the same sources with the measurement
    variables of tol_amd/csrc/knobs.h compiled in.  A/B tests that force a launch form build their objects with
    `library=measure`; the shipped library ignores those variables."""
    return tolfg.measure_lib()
