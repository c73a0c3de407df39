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


def load_golden(name):
    arrays = dict(np.load(os.path.join(GOLDEN, f"{name}.npz")))
    with open(os.path.join(GOLDEN, f"{name}.json")) as f:
        meta = json.load(f)
    return arrays, meta


@pytest.fixture(scope="session")
def golden():
    return load_golden
