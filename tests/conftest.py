import os
import sys

import pytest

# the test process is the application here: every lane of `get` on a hardware queue of its own (the library itself leaves
# the environment alone); must happen before the HIP runtime initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "slow: long CPU test")
