import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden", "convonet_golden.npz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return dict(np.load(GOLDEN))


@pytest.fixture(scope="session")
def np_weights():
    from oracle import convonet_oracle as O
    return O.make_random_weights(0)


@pytest.fixture(scope="session")
def oracle_weights(np_weights):
    from oracle import convonet_oracle as O
    return O.to_torch(np_weights)
