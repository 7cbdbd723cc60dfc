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


def load_acquisitions():
    """The reference's acquisition layer, verbatim: GPyOpt.acquisitions.{base,EI,LCB,MPI,LP} and
    GPyOpt.core.evaluators.batch_local_penalization (estimate_L, LocalPenalization).

    Their import chain reaches ``GPyOpt.models`` (-> GPy -> paramz, absent) only for the NAME ``GPModel`` that
    core/task/cost.py binds at import time and never touches on this path (constant cost); an empty placeholder
    module supplies that name, every other module on the chain is the reference's own file.  Returns a dict of modules.
    """
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF)
    load()
    _pkg("GPyOpt.core.task", REF + "/GPyOpt/GPyOpt/core/task")
    _pkg("GPyOpt.core.evaluators", REF + "/GPyOpt/GPyOpt/core/evaluators")
    _pkg("GPyOpt.acquisitions", REF + "/GPyOpt/GPyOpt/acquisitions")
    if "GPyOpt.models" not in sys.modules:
        m = types.ModuleType("GPyOpt.models")
        m.GPModel = type("GPModel", (object,), {})   # name only; see the docstring
        sys.modules["GPyOpt.models"] = m
    names = {"base": "GPyOpt.acquisitions.base", "EI": "GPyOpt.acquisitions.EI", "LCB": "GPyOpt.acquisitions.LCB",
             "MPI": "GPyOpt.acquisitions.MPI", "LP": "GPyOpt.acquisitions.LP",
             "lp_evaluator": "GPyOpt.core.evaluators.batch_local_penalization"}
    mods = {k: importlib.import_module(v) for k, v in names.items()}
    # compute_batch does ``from ...acquisitions import AcquisitionLP`` at call time
    sys.modules["GPyOpt.acquisitions"].AcquisitionLP = mods["LP"].AcquisitionLP
    return mods


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
