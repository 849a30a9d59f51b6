import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_iou():
    return np.load(os.path.join(GOLDEN, "iou_nms.npz"))


@pytest.fixture(scope="session")
def golden_geom():
    return np.load(os.path.join(GOLDEN, "geometry.npz"))


@pytest.fixture(scope="session")
def golden_core():
    return np.load(os.path.join(GOLDEN, "core.npz"), allow_pickle=True)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O
