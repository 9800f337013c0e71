"""PCA update / incremental PCA checks shared by the CPU tier (fake library) and the GPU tier (librlhip.so):
raleigh_amd.interfaces.pca(have=..., batch_size=...) against the reference's figures for the same seeded data
(tests/golden/known_answers.json, made by tests/golden/make_golden.py from raleigh/interfaces/pca.py:142-164 ->
lra.py:157-425) and against properties the result must have whatever the path: orthonormal components,
orthogonal reduced data in descending order, the exact mean of ALL rows, the error the tolerance asks for."""

import json
import os

import numpy as np


def known(golden_dir):
    return json.load(open(os.path.join(golden_dir, 'known_answers.json')))


def data_600x400():
    from oracle.pca_data import generate
    np.random.seed(1)
    A, sigma, u, v = generate(600, 400, 200, pca=True)
    return A


def check_shape_of_result(A, mean, trans, comps, ortho_tol=2e-4):
    k = comps.shape[0]
    assert mean.shape == (1, A.shape[1]) and trans.shape == (A.shape[0], k) and comps.shape == (k, A.shape[1])
    assert np.allclose(mean, A.mean(axis=0, keepdims=True), atol=1e-6)
    assert np.abs(comps @ comps.T - np.eye(k)).max() < ortho_tol
    G = (trans.T @ trans).astype(np.float64)
    d = np.diag(G)
    assert np.abs(G - np.diag(d)).max() < 1e-4 * d[0]
    assert np.all(np.diff(d) <= 1e-5 * d[0])


def update_with_tolerance(golden_dir):
    from raleigh_amd.interfaces import pca, pca_error
    k = known(golden_dir)['pca_600x400_update_tol']
    A = data_600x400()
    A0, A1 = A[:480], A[480:]
    mean, trans, comps = pca(A0, tol=0.05)
    mean, trans, comps = pca(A1, have=(mean, trans, comps))          # fewer new rows than columns
    check_shape_of_result(A, mean, trans, comps)
    em, ef = pca_error(A, mean, trans, comps)
    assert ef <= 1.05 * k["ef"] and em <= 1.2 * k["em"]
    assert abs(comps.shape[0] - k['ncomp']) <= 0.1 * k['ncomp']
    sv = np.linalg.norm(trans, axis=0)[:10]
    assert np.max(np.abs(sv - np.array(k['sigma'])) / k['sigma'][0]) < 2e-3


def update_keeps_the_number_of_components(golden_dir):
    from raleigh_amd.interfaces import pca, pca_error
    k = known(golden_dir)['pca_600x400_update_npc30']
    A = data_600x400()
    A0, A1 = A[:480], A[480:]
    mean, trans, comps = pca(A0, npc=30)
    mean, trans, comps = pca(A1, have=(mean, trans, comps))
    # "the same number of components as in R0" (pca.py:52-57); the reference itself returns 33 here
    assert comps.shape[0] == 30
    check_shape_of_result(A, mean, trans, comps)
    em, ef = pca_error(A, mean, trans, comps)
    assert ef <= 1.05 * k['ef'] and em <= 1.2 * k['em']
    exact = np.linalg.svd((A - A.mean(axis=0)).astype(np.float64), compute_uv=False)[:30]
    sv = np.linalg.norm(trans, axis=0)
    assert np.max(np.abs(sv[:10] - exact[:10]) / exact[0]) < 2e-3


def incremental(golden_dir):
    from raleigh_amd.interfaces import pca, pca_error
    kn = known(golden_dir)
    A = data_600x400()
    mean, trans, comps = pca(A, batch_size=200, tol=0.05)            # batches of fewer rows than columns
    check_shape_of_result(A, mean, trans, comps)
    em, ef = pca_error(A, mean, trans, comps)
    k = kn['pca_600x400_incremental_tol']
    assert ef <= 0.05 and ef <= 1.05 * k['ef'] and em <= 1.2 * k['em']
    mean, trans, comps = pca(A, batch_size=200, npc=30)
    assert comps.shape[0] == 30
    check_shape_of_result(A, mean, trans, comps)
    em, ef = pca_error(A, mean, trans, comps)
    k = kn['pca_600x400_incremental_npc30']
    assert ef <= 1.05 * k['ef'] and em <= 1.2 * k['em']


def tall_batches():
    """More rows than columns in every batch (the eigenvectors of the deflated operator live in the column
    space), a cap on the number of components, and an update whose new rows the components in hand
    already describe."""
    from raleigh_amd.interfaces import pca, pca_error
    from oracle.pca_data import generate
    np.random.seed(1)
    A, sigma, u, v = generate(1200, 300, 150, pca=True)
    mean, trans, comps = pca(A, batch_size=400, tol=0.05)
    check_shape_of_result(A, mean, trans, comps)
    em, ef = pca_error(A, mean, trans, comps)
    assert ef <= 0.05 * 1.1
    full = pca(A, tol=0.05)[2].shape[0]
    assert full <= comps.shape[0] <= 1.6 * full
    mean, trans, comps = pca(A, batch_size=600, tol=0.05, mpc=12)
    assert comps.shape[0] <= 12
    check_shape_of_result(A, mean, trans, comps)
    # rows that repeat old ones: nothing new to find, the mean and the reduced data still follow
    mean, trans, comps = pca(A[:600], npc=149)
    mean, trans, comps = pca(np.ascontiguousarray(A[:50]), have=(mean, trans, comps))
    B = np.concatenate((A[:600], A[:50]))
    check_shape_of_result(B, mean, trans, comps)
    em, ef = pca_error(B, mean, trans, comps)
    assert ef < 2e-2


def refusals():
    import pytest
    from raleigh_amd.interfaces import pca
    from raleigh_amd.interfaces.lra import LowerRankApproximation
    from raleigh_amd.algebra.dense_matrix import AMatrix
    A = data_600x400()
    mean, trans, comps = pca(A[:300], npc=5)
    with pytest.raises(ValueError):
        pca(np.ascontiguousarray(A[300:, :399]), have=(mean, trans, comps))       # another number of columns
    with pytest.raises(ValueError):
        pca(A[300:].astype(np.float64), have=(mean, trans, comps))                # another data type
    with pytest.raises(ValueError):
        pca(A[300:], have=(mean, trans[:, :4], comps))                            # trans / comps disagree
    with pytest.raises(RuntimeError):
        LowerRankApproximation().update(AMatrix(A[300:]))                         # nothing to update
    with pytest.raises(ValueError):
        pca(A, batch_size=100, npc=5, norm='x')


def update_with_other_norms(golden_dir):
    """pca(have=...) stopped by the largest row ('m': against the reference's own result) and by the largest singular
    value ('s': the reference fails there with an IndexError -- lra.py:342 indexes the singular values of the OLD
    approximation with the new number of components -- so the property itself is checked) of the remainder; then the
    incremental variant under 'm'."""
    from raleigh_amd.interfaces import pca, pca_error
    k = known(golden_dir)['pca_600x400_update_tol_m']
    assert 'failed' in known(golden_dir)['pca_600x400_update_tol_s']
    A = data_600x400()
    A0, A1 = A[:480], A[480:]
    As = (A - A.mean(axis=0)).astype(np.float64)
    mean, trans, comps = pca(A0, tol=0.05, norm='m')
    assert abs(comps.shape[0] - k['ncomp_before']) <= 0.1 * k['ncomp_before']
    mean, trans, comps = pca(A1, have=(mean, trans, comps), tol=0.05, norm='m')
    check_shape_of_result(A, mean, trans, comps)
    em, ef = pca_error(A, mean, trans, comps)
    assert em <= 0.05 * 1.05 and em <= 1.5 * k['em'] and ef <= 1.2 * k['ef']
    assert abs(comps.shape[0] - k['ncomp']) <= 0.15 * k['ncomp']
    sv = np.linalg.norm(trans, axis=0)[:10]
    assert np.max(np.abs(sv - np.array(k['sigma'])) / k['sigma'][0]) < 2e-3
    mean, trans, comps = pca(A0, tol=0.05, norm='s')
    mean, trans, comps = pca(A1, have=(mean, trans, comps), tol=0.05, norm='s')
    check_shape_of_result(A, mean, trans, comps)
    D = (trans @ comps).astype(np.float64) - As
    assert np.linalg.norm(D, 2) <= 0.05 * np.linalg.norm(As, 2) * 1.3     # (two truncations of tol each, in quadrature)
    mean, trans, comps = pca(A, batch_size=200, tol=0.05, norm='m')
    check_shape_of_result(A, mean, trans, comps)
    em, ef = pca_error(A, mean, trans, comps)
    assert em <= 0.05 * 1.5


def fewer_samples_than_features():
    """One-shot PCA of 300 x 700 fp32 rows: components orthonormal (the refinement of pca.py:146-147), reduced data
    orthogonal and descending, error as asked."""
    from raleigh_amd.interfaces import pca, pca_error
    from oracle.pca_data import generate
    np.random.seed(1)
    A, sigma, u, v = generate(300, 700, 150, pca=True)
    for kw in (dict(npc=20), dict(tol=0.1)):
        mean, trans, comps = pca(A, **kw)
        check_shape_of_result(A, mean, trans, comps, ortho_tol=1e-5)
        em, ef = pca_error(A, mean, trans, comps)
        assert ef <= (0.1 * 1.02 if 'tol' in kw else 0.12)


def other_norms():
    """One-shot PCA stopped by the largest row norm ('m') or the largest singular value ('s') of the remainder
    (lra.py:109-149 -> truncated_svd.py:206-283 with the mean shift), both shapes of data."""
    from raleigh_amd.interfaces import pca
    from oracle.pca_data import generate
    for (m, n) in ((600, 400), (300, 700)):
        np.random.seed(1)
        A, sigma, u, v = generate(m, n, 150, pca=True)
        As = A - A.mean(axis=0)
        smax = np.linalg.norm(As.astype(np.float64), 2)
        rows = lambda D: np.sqrt((D * D).sum(1).max())
        mean, trans, comps = pca(A, tol=0.05, norm='m')
        assert rows(trans @ comps - As) <= 0.05 * rows(As) * 1.01
        k_m = comps.shape[0]
        mean, trans, comps = pca(A, tol=0.05, norm='s')
        assert np.linalg.norm((trans @ comps - As).astype(np.float64), 2) <= 0.05 * smax * 1.01
        mean, trans, comps = pca(A, tol=0.001, norm='m', mpc=10)
        assert comps.shape[0] == 10 and k_m > 10
