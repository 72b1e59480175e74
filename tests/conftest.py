import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
