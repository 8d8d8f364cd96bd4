import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def small_problem():
    """K=4, 150 corr/pair, background sphere (100 % valid pixels), full-resolution frames."""
    from bundletrack_amd import synthetic as S
    return S.make_problem(4, 150, seed=3, background=True)


@pytest.fixture(scope="session")
def small_problem_masked():
    """K=4, 150 corr/pair, object only (~5 % valid pixels)."""
    from bundletrack_amd import synthetic as S
    return S.make_problem(4, 150, seed=4, background=False)
