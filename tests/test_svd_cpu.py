"""CPU tier: Vectors.svd() on rank-deficient and ill-conditioned blocks (host logic over tests/fake_lib.py),
cases of tests/_svd_cases.py."""

import pytest

import fake_lib
import _svd_cases as cases


@pytest.fixture(autouse=True)
def fake():
    f = fake_lib.install()
    yield f
    fake_lib.uninstall()


@pytest.mark.parametrize('dt,cond,rank', cases.CASES)
def test_svd_survives_rank_loss(dt, cond, rank):
    cases.check(dt, cond, rank, n=600, m=16)
