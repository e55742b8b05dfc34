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


class Golden:
    """Read-only view of one fixture file with '/'-separated keys."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    def __getitem__(self, key):
        v = self.z[key]
        return v.item() if v.shape == () else v

    def has(self, key):
        return key in self.z.files

    def group(self, prefix):
        p = prefix.rstrip("/") + "/"
        return {k[len(p):]: self[k] for k in self.z.files if k.startswith(p) and "/" not in k[len(p):]}

    def cases(self, depth=1):
        return sorted({"/".join(k.split("/")[:depth]) for k in self.z.files})


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get
