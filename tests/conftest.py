import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _gpu_present():
    # device_count() does not initialise the HIP runtime on this image (is_available() does); the
    # multi-process tests rely on this process not having touched the GPU when it starts helpers
    import torch
    return torch.cuda.device_count() > 0


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Worker processes of the multi-process tests are forked from a fork server that is started
    # HERE, before anything in this process can have initialised the GPU: a process that has must
    # never exec another program on the GPU pool, and a fork server never execs for its children.
    import multiprocessing as mp
    from multiprocessing import forkserver
    mp.set_forkserver_preload(["numpy"])
    forkserver.ensure_running()


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
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
