import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def weights_denoiser():
    import nhans_amd  # noqa: F401
    from nhans_amd import weights
    return weights.synthetic_weights("denoiser", 7)


@pytest.fixture(scope="session")
def weights_separator():
    import nhans_amd  # noqa: F401
    from nhans_amd import weights
    return weights.synthetic_weights("separator", 7)


@pytest.fixture(scope="session")
def lib_built():
    """Build the HIP library once per session (no-op when up to date)."""
    import __graft_entry__ as g
    g.build()
    return True


def load_case(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))
