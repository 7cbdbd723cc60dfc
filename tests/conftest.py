import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "gp_golden.npz"))


def golden_tags(g):
    return sorted({k.split("/")[0] for k in g.files if k.startswith("N")})


class Case(object):
    def __init__(self, g, tag):
        self._g, self._t = g, tag

    def __getattr__(self, k):
        return self._g[self._t + "/" + k]


def relmax(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
