import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """The plain-C oracle (oracle/liboracle.so); built on demand with gcc (checker only)."""
    import ctypes
    import subprocess
    d = os.path.join(ROOT, "oracle")
    so = os.path.join(d, "liboracle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", d], check=True, capture_output=True)
    return ctypes.CDLL(so)
