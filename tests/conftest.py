import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden", "convonet_golden.npz")


PRECISION_MODES = ("f32", "bf16x6")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "both_precisions: run the test once per arithmetic of the decoder's dense layers - f32 MFMA and the "
                                       "f32-equivalent bf16x6 split (ifd_opt_params.precision) - at the SAME bars (round-5 verdict, item 1)")


def pytest_generate_tests(metafunc):
    if metafunc.definition.get_closest_marker("both_precisions") is not None:
        assert "precision_mode" in metafunc.fixturenames, metafunc.function.__name__
        metafunc.parametrize("precision_mode", PRECISION_MODES, indirect=True)


@pytest.fixture
def precision_mode(request):
    """The arithmetic every optimise call of the test runs in unless it names one: sets the runtime's default precision (what
    `precision=None` means in Restorer.optimize_points / mesh_sample) for the duration of the test and hands the name to the test
    for the places that take it explicitly (DefenseArgs(precision=...), CLI flags)."""
    mode = getattr(request, "param", "f32")
    from ifdefense_amd import runtime
    prev = runtime.set_default_precision(mode)
    print("[precision_mode = %s]" % mode)
    yield mode
    runtime.set_default_precision(prev)


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
