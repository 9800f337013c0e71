"""truncated_svd checks shared by the CPU tier (fake library) and the GPU tier: singular values against the
reference's (tests/golden/known_answers.json 'tsvd_*', made by tests/golden/make_golden.py --truncated-svd-only
from raleigh/interfaces/truncated_svd.py) and the exact SVD, A V = U S with orthonormal factors, and each norm
of the truncation error within the tolerance asked for."""

import json
import os

import numpy as np


def _case(m, n, dt, rank=200):
    from oracle.pca_data import generate
    np.random.seed(1)
    A, s, uu, vv = generate(m, n, rank, dtype=dt)
    return A


def _check_factors(A, u, sg, vt, tol):
    k = len(sg)
    assert u.shape == (A.shape[0], k) and vt.shape == (k, A.shape[1])
    assert np.all(np.diff(sg) <= 0)
    assert np.abs(u.T @ u - np.eye(k)).max() < tol and np.abs(vt @ vt.T - np.eye(k)).max() < tol
    # the factor not orthonormalised last carries the solver's vector tolerance, sqrt(eps) (truncated_svd.py:96-99)
    assert np.linalg.norm(A @ vt.T - u * sg) <= 10 * np.sqrt(np.finfo(A.dtype).eps) * sg[0]


def run(golden_dir, m, n, dt):
    from raleigh_amd.interfaces import truncated_svd
    known = json.load(open(os.path.join(golden_dir, 'known_answers.json')))['tsvd_%dx%d' % (m, n)]
    A = _case(m, n, dt)
    single = dt == np.float32
    exact = np.linalg.svd(A.astype(np.float64), compute_uv=False)
    u, sg, vt = truncated_svd(A, nsv=20)
    assert 20 <= len(sg) <= 2 * known['ncomp_nsv20']
    _check_factors(A, u, sg, vt, 5e-6 if single else 1e-11)
    ref = np.array(known['sigma_nsv20'])
    assert np.max(np.abs(sg[:20] - ref)) <= (2e-6 if single else 1e-12) * ref[0]
    assert np.max(np.abs(sg - exact[:len(sg)])) <= (2e-6 if single else 1e-12) * exact[0]
    for name, kw, pos in (('s', dict(tol=0.1, norm='s'), 0), ('f', dict(tol=0.1, norm='f'), 1), ('m', dict(tol=0.2, norm='m'), 2)):
        u, sg, vt = truncated_svd(A, **kw)
        _check_factors(A, u, sg, vt, 5e-6 if single else 1e-11)
        D = A - (u * sg) @ vt
        err = (np.linalg.norm(D, 2) / np.linalg.norm(A, 2), np.linalg.norm(D) / np.linalg.norm(A),
               np.sqrt((D * D).sum(1).max() / (A * A).sum(1).max()))[pos]
        assert err <= kw['tol'] * 1.001
        # no fewer than the exact SVD needs, no more than twice what the reference took
        if name == 's':
            need = int(np.argmax(exact <= 0.1 * exact[0]))
        elif name == 'f':
            tail = np.sqrt(np.cumsum((exact ** 2)[::-1])[::-1])
            need = int(np.argmax(tail <= 0.1 * tail[0]))
        else:
            need = 1
        assert need <= len(sg) <= max(2 * known['ncomp_' + name], need + 128)
    u, sg, vt = truncated_svd(A, tol=0.01, norm='f', msv=15)
    assert len(sg) == 15
    _check_factors(A, u, sg, vt, 5e-6 if single else 1e-11)


def refusals():
    import pytest
    from raleigh_amd.interfaces import truncated_svd
    A = _case(60, 40, np.float64, rank=20)
    with pytest.raises(ValueError):
        truncated_svd(A, nsv=3, norm='x')
    with pytest.raises(ValueError):
        truncated_svd(A)                       # neither nsv nor tol: the interactive mode is not offered
    with pytest.raises(ValueError):
        truncated_svd(A[0], nsv=1)
