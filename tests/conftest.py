import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _blas_threads_match_the_cpu_share():
    """The oracle's LAPACK legs (full-size cases) run with as many BLAS threads as this job may use: a GPU box shows
    64+ cores to os.cpu_count() while the job's share is 16, and an oversubscribed dpotrf is 5x slower."""
    from oracle import cpu_ref as O
    lim, _ = O.limit_blas_threads()
    yield
    del lim


def emulation_modes():
    """Both arithmetic modes of the device path for the driver-run parity suite: true fp64 (the default, the headline) and
    the int8 residue emulation of the bulk contractions (option "emulate_fp64", csrc/rns.hip)."""
    return [pytest.param(0, id="fp64"), pytest.param(1, id="emulated")]


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
