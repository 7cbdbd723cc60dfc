"""CPU oracle for the exact-GP hot path -- TEST INFRASTRUCTURE ONLY.

This package is a NumPy/SciPy restatement of the reference's algorithm
(GPy 1.9.6 + GPyOpt 1.2.5 as vendored under /root/reference).  It exists so
that tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg can
check / time the HIP path against the reference's arithmetic.  Nothing under
``gaussian_process_optimization_amd/`` may import it: the product path has no
CPU fallback and fails loudly when the HIP library is missing.

Parity status: PINNED.  ``oracle/pin_against_reference.py`` (run in the build
container, where /root/reference is mounted) checks every function here
against the verbatim-imported reference leaf modules ``GPy.util.linalg``,
``GPy.util.diag``, ``GPy.util.normalizer``, ``GPyOpt.util.general`` and the
reference's own ``stationary_utils.c`` compiled into ``oracle/_ref/``; the
golden vectors in tests/golden/ were produced by
``tests/golden/generate_golden.py`` through those same reference modules.
The whole GPy/GPyOpt packages cannot be imported (``paramz`` is an absent,
un-vendored dependency), so the Python callers that need paramz
(Stationary/RBF/Matern52, ExactGaussianInference, PosteriorExact, GP.predict,
GPModel, Acquisition*) are restated line by line with file:line citations.
"""
