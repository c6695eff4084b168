import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


REF_PRESENT = os.path.isfile("/root/reference/src/inverse_compositional_algorithm.py")
needs_reference = pytest.mark.skipif(not REF_PRESENT, reason="/root/reference not present (GPU box)")
