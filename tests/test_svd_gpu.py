"""GPU tier: Vectors.svd() on rank-deficient and ill-conditioned blocks through librlhip.so, cases of tests/_svd_cases.py."""

import pytest

import _svd_cases as cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dt,cond,rank', cases.CASES)
def test_svd_survives_rank_loss(dt, cond, rank):
    cases.check(dt, cond, rank)


def test_svd_wide_block():
    import numpy as np
    cases.check(np.float64, 1e12, None, n=200000, m=64)
