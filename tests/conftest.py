import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def port_oracle():
    from oracle import oracle as O
    return O.PortOracle()


@pytest.fixture(scope="session")
def ref_oracle():
    from oracle import oracle as O
    if not O.have_reference():
        pytest.skip("oracle/_ref not built (needs /root/reference in the build container)")
    return O.ReferenceOracle()
