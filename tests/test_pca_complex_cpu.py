"""CPU tier: complex PCA with the mean shift (host logic over tests/fake_lib.py), cases of tests/_pca_complex_cases.py."""

import numpy as np
import pytest

import fake_lib
import _pca_complex_cases as cases


@pytest.fixture(autouse=True)
def fake():
    f = fake_lib.install()
    yield f
    fake_lib.uninstall()


SHAPES = [(120, 40, np.complex128), (40, 120, np.complex128), (120, 40, np.complex64), (40, 120, np.complex64)]


@pytest.mark.parametrize('m,n,dt', SHAPES)
def test_operator_against_numpy(m, n, dt):
    cases.operator_against_numpy(m, n, dt)


@pytest.mark.parametrize('m,n,dt', SHAPES)
def test_pca_is_optimal(m, n, dt):
    cases.pca_is_optimal(m, n, dt)


@pytest.mark.parametrize('m,n,dt', [(400, 160, np.complex128), (160, 400, np.complex64)])
def test_row_norm_rule_with_shift(m, n, dt):
    cases.row_norm_rule_with_shift(m, n, dt)


@pytest.mark.parametrize('m0,m1,n,dt', [(300, 120, 90, np.complex128), (260, 60, 80, np.complex64)])
def test_update_matches_one_shot(m0, m1, n, dt):
    cases.update_matches_one_shot(m0, m1, n, dt)
