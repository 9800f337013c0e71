"""CPU tier: PCA update and incremental PCA (host logic of raleigh_amd/interfaces/lra.py over tests/fake_lib.py)
against the reference's own figures and the properties of tests/_pca_update_cases.py."""

import numpy as np
import pytest

import fake_lib
import _pca_update_cases as cases


@pytest.fixture(autouse=True)
def fake():
    f = fake_lib.install()
    yield f
    fake_lib.uninstall()


def test_update_with_tolerance(golden_dir):
    cases.update_with_tolerance(golden_dir)


def test_update_keeps_the_number_of_components(golden_dir):
    cases.update_keeps_the_number_of_components(golden_dir)


def test_incremental(golden_dir):
    cases.incremental(golden_dir)


def test_tall_batches():
    cases.tall_batches()


def test_refusals():
    cases.refusals()


def test_fewer_samples_than_features():
    cases.fewer_samples_than_features()


def test_other_norms():
    cases.other_norms()


def test_update_with_other_norms(golden_dir):
    cases.update_with_other_norms(golden_dir)


def test_fp32_batch_with_as_many_components_as_rows():
    """200 x 400 fp32 rows, tolerance mode: the block solver hands the whole problem to the dense
    Rayleigh-Ritz step in the complement (solver.py:502-585).  With the random basis orthonormalised in
    double precision the approximation is exact to rounding; rotated in fp32 with "dependent" directions
    dropped (what the reference does) sigma_max comes out 1.6 % low."""
    from raleigh_amd.interfaces import pca, pca_error
    A = cases.data_600x400()[:200]
    np.random.seed(3)
    mean, trans, comps = pca(A, tol=0.05)
    em, ef = pca_error(A, mean, trans, comps)
    assert ef < 1e-4
    exact = np.linalg.svd((A - A.mean(axis=0)).astype(np.float64), compute_uv=False)
    sv = np.linalg.norm(trans, axis=0)
    assert np.max(np.abs(sv[:20] - exact[:20])) < 1e-5 * exact[0]
