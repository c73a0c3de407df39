"""pytest configuration: markers, import paths, shared fixtures."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "myrtle-vision_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """The training-loop tests (tests/test_train_gpu.py) start DataLoader worker processes by fork.  Forking this process late in
    a session -- after the kernel and parity tests have mapped tens of GB -- costs seconds per worker: the same eight tests took
    445 s at the end of the GPU suite and 45 s on their own (measured, round 4).  They run first; everything else keeps its order."""
    first = [it for it in items if it.fspath.basename == "test_train_gpu.py"]
    if first and len(first) != len(items):
        rest = [it for it in items if it.fspath.basename != "test_train_gpu.py"]
        items[:] = first + rest


def load_golden(name):
    arrays = dict(np.load(os.path.join(GOLDEN, f"{name}.npz")))
    with open(os.path.join(GOLDEN, f"{name}.json")) as f:
        meta = json.load(f)
    return arrays, meta


@pytest.fixture(scope="session")
def golden():
    return load_golden
