import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden():
    """name -> dict of arrays frozen from the unmodified reference (tests/golden/make_golden.py)."""
    cache = {}

    def load(name):
        if name not in cache:
            z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
            cache[name] = {k: z[k] for k in z.files}
        return cache[name]

    return load


def rms_error(a, b):
    d = np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)
    return float(np.sqrt(np.mean(d * d)))
