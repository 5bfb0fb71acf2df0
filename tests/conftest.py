import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def port_oracle():
    from oracle.loader import PortOracle, build
    build()
    return PortOracle()


@pytest.fixture(scope="session")
def ref_oracle():
    from oracle.loader import RefOracle, ref_available
    if not ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference in the build container)")
    return RefOracle()
