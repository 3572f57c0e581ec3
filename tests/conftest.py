import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no GPU is visible and they were not deselected."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import json
    import numpy as np

    class G:
        manifest = json.load(open(os.path.join(GOLDEN, "manifest.json")))

        @staticmethod
        def load(name):
            return np.load(os.path.join(GOLDEN, name + ".npz"))
    return G


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; builds oracle/libgraspbal_oracle.so on first use)."""
    from oracle import oracle
    oracle.build()
    return oracle
