"""GPU tier: truncated_svd through librlhip.so, cases of tests/_truncated_svd_cases.py."""

import numpy as np
import pytest

import _truncated_svd_cases as cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('m,n,dt', [(600, 400, np.float32), (300, 700, np.float64)])
def test_truncated_svd(golden_dir, m, n, dt):
    cases.run(golden_dir, m, n, dt)


def test_refusals():
    cases.refusals()
