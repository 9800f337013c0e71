"""CPU tier (host logic over tests/fake_lib.py): generalized and buckling problems, tests/_generalized_cases.py."""

import pytest

import fake_lib
import _generalized_cases as cases


@pytest.fixture(autouse=True)
def fake():
    f = fake_lib.install()
    yield f
    fake_lib.uninstall()


def test_generalized_preconditioned(golden_dir):
    cases.generalized_preconditioned(golden_dir)


def test_reference_generalized_mode_is_a_product(golden_dir):
    cases.reference_generalized_mode_is_a_product(golden_dir)


def test_generalized_shift_invert(golden_dir):
    cases.generalized_shift_invert(golden_dir)


def test_buckling(golden_dir):
    cases.buckling(golden_dir)
