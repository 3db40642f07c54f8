import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_s0():
    import numpy as np
    return np.load(os.path.join(REPO, "tests", "golden", "s0_small.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_s1():
    import numpy as np
    return np.load(os.path.join(REPO, "tests", "golden", "s1_full.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def native():
    from oracle import native as nat
    nat.build(ref=True)
    return nat
