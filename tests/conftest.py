import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dusty-gan-v2_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected explicitly with -m gpu; skip them when no device is visible.
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name))
    return {k: torch.from_numpy(d[k]) for k in d.files}


def sub_dict(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def g_ops():
    return load_golden("ops.npz")


@pytest.fixture(scope="session")
def g_coords():
    return load_golden("coords.npz")


@pytest.fixture(scope="session")
def g_geometry():
    return load_golden("geometry.npz")


@pytest.fixture(scope="session")
def g_small():
    return load_golden("model_small.npz")


@pytest.fixture(scope="session")
def g_full():
    return load_golden("model_full.npz")
