import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; without a GPU they are skipped rather than failed.
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        with np.load(os.path.join(GOLD, name + ".npz")) as z:
            return {k: z[k] for k in z.files}
    return load


@pytest.fixture(scope="session")
def lib():
    """Build (if stale) and load libs2vt_hip.so."""
    from s2vt_video_caption_amd import build, capi
    build.build()
    return capi.load()
