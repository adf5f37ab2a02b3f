import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """ctypes handle on the CPU oracle (oracle/libdvt_oracle.so); built on demand."""
    from tests import _orc

    return _orc.load()


@pytest.fixture(scope="session")
def prover_lib():
    """ctypes handle on the product C-ABI library (HIP). Never falls back."""
    from dvt_circuits_amd import capi

    return capi.load()
