import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _exact_beam_mode():
    """The parity tests are written against the exact sum over all lattice arcs (the oracle without a beam); the
    drop-in's default is the reference's own semantics, output_beam pruning, which tests/test_pruned.py and the
    output_beam tests of test_gpu_parity.py select explicitly (and test_host.py pins as the default)."""
    import waterfall_amd as wf
    wf.set_output_beam_mode("exact")
    yield
    wf.set_output_beam_mode(wf.graph.DEFAULT_BEAM_MODE)


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
