"""CPU tier: truncated_svd (host logic over tests/fake_lib.py), cases of tests/_truncated_svd_cases.py."""

import numpy as np
import pytest

import fake_lib
import _truncated_svd_cases as cases


@pytest.fixture(autouse=True)
def fake():
    f = fake_lib.install()
    yield f
    fake_lib.uninstall()


@pytest.mark.parametrize('m,n,dt', [(600, 400, np.float32), (300, 700, np.float64)])
def test_truncated_svd(golden_dir, m, n, dt):
    cases.run(golden_dir, m, n, dt)


def test_refusals():
    cases.refusals()
