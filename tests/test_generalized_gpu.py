"""GPU tier: generalized and buckling problems on the real library, tests/_generalized_cases.py."""

import pytest

import _generalized_cases as cases

pytestmark = pytest.mark.gpu


def test_generalized_preconditioned(golden_dir):
    cases.generalized_preconditioned(golden_dir)


def test_reference_generalized_mode_is_a_product(golden_dir):
    cases.reference_generalized_mode_is_a_product(golden_dir)


def test_generalized_shift_invert(golden_dir):
    cases.generalized_shift_invert(golden_dir)


def test_buckling(golden_dir):
    cases.buckling(golden_dir)
