"""Loader for the reference's importable leaf modules (BUILD CONTAINER ONLY).

``import GPy`` fails (paramz is absent), but a few leaf modules have no such
dependency.  They are imported verbatim from /root/reference by registering
empty parent packages so that ``GPy/__init__.py`` never runs.  Used by
oracle/pin_against_reference.py and tests/golden/generate_golden.py; never on
the GPU box (the reference does not travel) and never by the product.
"""
import ctypes
import importlib
import os
import sys
import types

REF = os.environ.get("GP_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    return os.path.isdir(os.path.join(REF, "GPy", "GPy", "util"))


def _pkg(name, path):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m


def load():
    """Returns (linalg, diag, normalizer, general) reference modules."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF)
    _pkg("GPy", REF + "/GPy/GPy")
    _pkg("GPy.util", REF + "/GPy/GPy/util")
    _pkg("GPyOpt", REF + "/GPyOpt/GPyOpt")
    _pkg("GPyOpt.util", REF + "/GPyOpt/GPyOpt/util")
    _pkg("GPyOpt.core", REF + "/GPyOpt/GPyOpt/core")
    linalg = importlib.import_module("GPy.util.linalg")
    diag = importlib.import_module("GPy.util.diag")
    normalizer = importlib.import_module("GPy.util.normalizer")
    general = importlib.import_module("GPyOpt.util.general")
    return linalg, diag, normalizer, general


def load_stationary_utils():
    """ctypes handle on oracle/_ref/libstationary_utils.so (reference C, compiled by oracle/Makefile)."""
    path = os.path.join(HERE, "_ref", "libstationary_utils.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    dp = ctypes.POINTER(ctypes.c_double)
    lib._grad_X.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, dp, dp, dp]
    lib._grad_X.restype = None
    lib._lengthscale_grads.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, dp, dp, dp]
    lib._lengthscale_grads.restype = None
    return lib
