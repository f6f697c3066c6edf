import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "multigpu: needs a node with SEVERAL MI355X (run with -m multigpu there; never part of -m gpu: the "
                                       "build sessions' boxes have one GPU, so these tests have not run yet)")


def pytest_collection_modifyitems(config, items):
    """The `multigpu` tests (tests/test_multi_real_devices.py) are collected only when asked for by name (`-m multigpu`): they belong
    neither to the CPU run (`-m "not gpu"`) nor to the one-GPU run (`-m gpu`)."""
    if "multigpu" in (config.getoption("-m") or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("multigpu") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


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
