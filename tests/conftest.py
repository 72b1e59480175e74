import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_sees_the_gpu_first(request):
    """On the GPU box, let torch initialise its HIP context before libspecscan.so creates streams, so both
    share the primary context whatever the test order."""
    if request.config.getoption("-m") == "gpu":
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    yield


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def ref_mod(oracle_mod):
    if not oracle_mod.have_ref():
        pytest.skip("oracle/_ref/libref_specscan.so not built (needs /root/reference)")
    oracle_mod.ref()
    return oracle_mod


@pytest.fixture(autouse=True)
def _whole_suite_on_the_diagnostics_build():
    """SS_TEST_USE_DIAG_LIB=1 (with SS_* variables of one's choice) runs every test against libspecscan_diag.so: an
    alternative implementation meets the whole suite, not only tests/test_gpu_parity.py's ALTERNATIVES."""
    if not os.environ.get("SS_TEST_USE_DIAG_LIB"):
        yield
        return
    import rtl_sdr_scanner_cpp_amd as pkg
    pkg.engine.use_diag_library(True)
    yield
    pkg.engine.use_diag_library(False)


@pytest.fixture
def diag_lib():
    """Engines created inside the test load libspecscan_diag.so — the -DSS_DIAG build, the only one that lets SS_* / SC_*
    environment variables select an alternative implementation (the product library never reads the environment)."""
    import rtl_sdr_scanner_cpp_amd as pkg
    pkg.engine.use_diag_library(True)
    yield
    pkg.engine.use_diag_library(False)
