"""Dual-threshold incomplete LU, ILUT(p, tau) (Y. Saad, Numer. Linear Algebra Appl. 1 (1994)), the
algorithm behind mkl dcsrilut that the reference's IncompleteLU calls
(raleigh/algebra/mkl_wrap.py:279-331: tol, maxfil = max_fill_rel * nnz / n).

TEST INFRASTRUCTURE ONLY: a pure-Python restatement for small matrices, the checker of the
library's host factorisation rlh_ilut_factor.  MKL itself is closed; its documented semantics
(drop entries below tol * ||row||_2, keep at most maxfil entries in the L part and in the U part
of every row, replace tiny pivots) are what is restated."""

import numpy as np
import scipy.sparse as sp


def ilut(a, tol, maxfil):
    a = sp.csr_matrix(a)
    a.sort_indices()
    n = a.shape[0]
    dt = np.complex128 if np.iscomplexobj(a.data) else np.float64
    lrows, urows = [], []             # lists of (cols, vals); urows with the diagonal first
    for i in range(n):
        cols = a.indices[a.indptr[i]:a.indptr[i + 1]]
        vals = a.data[a.indptr[i]:a.indptr[i + 1]].astype(dt)
        tau = tol * np.sqrt(np.sum(np.abs(vals) ** 2))
        w = {}
        for c, v in zip(cols, vals):
            w[int(c)] = w.get(int(c), 0) + v
        w.setdefault(i, 0)
        lower = []
        done = set()
        while True:
            cand = [k for k in w if k < i and k not in done]
            if not cand:
                break
            k = min(cand)
            done.add(k)
            ucols, uvals = urows[k]
            lik = w[k] / uvals[0]
            if abs(lik) < tau:
                del w[k]
                continue
            w[k] = lik
            lower.append(k)
            for c, v in zip(ucols[1:], uvals[1:]):
                w[int(c)] = w.get(int(c), 0) - lik * v

        def largest(keys):
            keys = sorted(keys, key=lambda j: (-abs(w[j]), j))[:maxfil]
            return sorted(keys)
        lk = largest(lower)
        uk = largest([j for j in w if j > i and abs(w[j]) >= tau])
        d = w[i]
        if abs(d) < tau or d == 0:
            d = tau if tau > 0 else 1e-4 * np.sqrt(np.sum(np.abs(vals) ** 2))
        lrows.append((lk, [w[j] for j in lk]))
        urows.append(([i] + uk, [d] + [w[j] for j in uk]))

    def assemble(rows):
        indptr = np.cumsum([0] + [len(c) for c, _ in rows])
        indices = np.array([c for cs, _ in rows for c in cs], dtype=np.int32)
        data = np.array([v for _, vs in rows for v in vs], dtype=dt)
        m = sp.csr_matrix((data, indices, indptr), shape=(n, n))
        m.sort_indices()
        return m
    return assemble(lrows), assemble(urows)
