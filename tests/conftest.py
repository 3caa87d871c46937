import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("OMP_NUM_THREADS", "4")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = ["A_numeric", "B_ragged", "M_mixed", "N_tf", "L_catonly", "K1_single"]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
