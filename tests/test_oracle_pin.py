"""CPU (build container only): the oracle against the reference's verbatim leaf modules and compiled C."""
import pytest

from oracle import pin_against_reference as pin
from oracle import ref_leaf

needs_ref = pytest.mark.skipif(not ref_leaf.available(), reason="reference tree not mounted (GPU box)")


@needs_ref
def test_linalg_bit_identical():
    assert "bit-identical" in pin.check_linalg(ref_leaf.load())


@needs_ref
def test_native_reductions():
    lib = ref_leaf.load_stationary_utils()
    if lib is None:
        pytest.skip("oracle/_ref not built")
    pin.check_native(lib)


@needs_ref
def test_lml_through_reference():
    pin.check_lml_through_reference(ref_leaf.load())


@needs_ref
def test_acquisition_layer_verbatim():
    """EI / LCB / MPI / LP and the local-penalisation evaluator: the reference's own modules on an oracle-backed model."""
    A = ref_leaf.load_acquisitions()
    assert "bit-identical" in pin.check_acquisitions(A)
    assert "compute_batch verbatim == oracle" in pin.check_lp_evaluator(A)


@needs_ref
def test_gower_space_and_table_loop_verbatim():
    """The reference's own Design_space and AcquisitionLP driving the run.py:1234-1258 loop under the Gower kernel."""
    A = ref_leaf.load_acquisitions()
    assert "table loop on verbatim AcquisitionLP == OracleLP" in pin.check_gower_space_and_table_loop(A)


def test_reference_test_invariants():
    # pinv closed form, var >= 0, normaliser equivalence, finite-difference gradients (no reference import needed)
    pin.check_invariants()
