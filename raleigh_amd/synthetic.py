"""Synthetic inputs of the BASELINE.json configurations: the matrices bench.py,
the GPU tests and the tools under tools/ run on.  Everything is a pure function of its arguments
(integer hashing, no RNG state, no libm calls in the values), so the GPU box and the build
container generate bit-identical matrices and committed reference eigenvalues stay valid.

* ``lap3d_rows``       rows [r0, r1) of the 7-point Laplacian of raleigh/examples/laplace.py:23-27
                       (config 1 / the roofline point) without forming the global matrix;
* ``hermitian_lap3d_rows``  the same plus i * skew first-neighbour coupling (config 5);
* ``fe_surrogate``     the stand-in for SuiteSparse shipsec5 (config 3: n = 179 860, ~56 nnz/row;
                       the .mtx is not available offline, README.md:20 of the reference only names it);
* ``read_matrix_market``  coordinate-format Matrix-Market reader, so that a supplied shipsec5.mtx
                       drops into the same tests.
"""

import gzip

import numpy as np
import scipy.sparse as sp


def lap3d_rows(nx, ny, nz, ax, ay, az, r0, r1):
    """Rows [r0, r1) of the 7-point Laplacian as a full (both triangles) CSR block with GLOBAL
    column indices."""
    n = nx * ny * nz
    r = np.arange(r0, r1, dtype=np.int64)
    ix, iy, iz = r % nx, (r // nx) % ny, r // (nx * ny)
    cx, cy, cz = ((nx + 1.0) / ax) ** 2, ((ny + 1.0) / ay) ** 2, ((nz + 1.0) / az) ** 2
    rows, cols, vals = [r], [r], [np.full(r.shape, 2 * (cx + cy + cz))]
    for cond, shift, c in ((ix > 0, -1, cx), (ix < nx - 1, 1, cx), (iy > 0, -nx, cy), (iy < ny - 1, nx, cy),
                           (iz > 0, -nx * ny, cz), (iz < nz - 1, nx * ny, cz)):
        rows.append(r[cond]); cols.append(r[cond] + shift); vals.append(np.full(int(cond.sum()), -c))
    rows = np.concatenate(rows) - r0
    blk = sp.csr_matrix((np.concatenate(vals), (rows, np.concatenate(cols))), shape=(r1 - r0, n))
    blk.sort_indices()
    return blk


def lap3d_coefficients(nx, ny, nz, ax, ay, az):
    return ((nx + 1.0) / ax) ** 2, ((ny + 1.0) / ay) ** 2, ((nz + 1.0) / az) ** 2


def hermitian_lap3d_rows(nx, ny, nz, ax, ay, az, r0, r1, skew=0.3, within_lines=True):
    """Rows [r0, r1) of H = L + i (S - S^T), L the 7-point Laplacian above and S = skew on the first
    superdiagonal: the complex Hermitian test operator of BASELINE config 5.

    within_lines=True keeps S inside the x-lines of the grid (no coupling between the last point
    of a line and the first of the next): H is then a Kronecker sum whose x-factor is a Hermitian
    tridiagonal Toeplitz matrix, and the spectrum is known in closed form
    (``hermitian_lap3d_eigenvalues``) -- the full-size runs are checked against it.
    within_lines=False is the global superdiagonal used by the small CPU-tier test."""
    n = nx * ny * nz
    L = lap3d_rows(nx, ny, nz, ax, ay, az, r0, r1).astype(np.complex128)
    r = np.arange(r0, r1, dtype=np.int64)
    ix = r % nx
    up = (ix < nx - 1) if within_lines else (r < n - 1)
    dn = (ix > 0) if within_lines else (r > 0)
    rows = np.concatenate((r[up], r[dn])) - r0
    cols = np.concatenate((r[up] + 1, r[dn] - 1))
    vals = np.concatenate((np.full(int(up.sum()), 1j * skew), np.full(int(dn.sum()), -1j * skew)))
    H = sp.csr_matrix(L + sp.csr_matrix((vals, (rows, cols)), shape=L.shape))
    H.sort_indices()
    return H


def hermitian_lap3d_eigenvalues(nx, ny, nz, ax, ay, az, skew=0.3):
    """All eigenvalues (sorted) of hermitian_lap3d_rows(..., within_lines=True): the x-factor
    tridiag(conj(b), 2 cx, b) with b = -cx + i skew has eigenvalues 2 cx - 2 |b| cos(k pi / (nx + 1))."""
    cx, cy, cz = lap3d_coefficients(nx, ny, nz, ax, ay, az)
    ex = 2 * cx - 2 * np.hypot(cx, skew) * np.cos(np.arange(1, nx + 1) * np.pi / (nx + 1))
    ey = 2 * cy - 2 * cy * np.cos(np.arange(1, ny + 1) * np.pi / (ny + 1))
    ez = 2 * cz - 2 * cz * np.cos(np.arange(1, nz + 1) * np.pi / (nz + 1))
    return np.sort((ex[:, None, None] + ey[None, :, None] + ez[None, None, :]).ravel())


def mass_matrix(nx, ny, nz):
    """A symmetric positive definite "mass" matrix on the nx x ny x nz grid of lap3d: the Kronecker product of the 1-D
    linear finite-element mass matrices tridiag(1/6, 2/3, 1/6) (27-point stencil): the B of the generalized problems
    A x = lambda B x of raleigh/interfaces/partial_hevp.py:103-200 in the tests."""
    def m1(k):
        return sp.diags([np.full(k - 1, 1.0 / 6), np.full(k, 2.0 / 3), np.full(k - 1, 1.0 / 6)], [-1, 0, 1], format='csr')
    b = sp.kron(m1(nz), sp.kron(m1(ny), m1(nx)), format='csr')
    b.sort_indices()
    return b


def stress_stiffness(nx, ny, nz, ax, ay, az, ratio=0.3):
    """A symmetric INDEFINITE "stress stiffness" matrix Ks on the grid of lap3d for the buckling problems
    (K + alpha Ks) v = 0 (raleigh/examples/buckling_evp.py): Ks = -(Dx - ratio Dy), Dx / Dy the second differences along x / y
    alone (compression along x, tension along y).  With K = lap3d(nx, ny, nz, ax, ay, az) everything commutes, so the load
    factors are known in closed form (`buckling_load_factors`)."""
    cx, cy, cz = lap3d_coefficients(nx, ny, nz, ax, ay, az)

    def d1(k, c):
        return sp.diags([np.full(k - 1, -c), np.full(k, 2 * c), np.full(k - 1, -c)], [-1, 0, 1], format='csr')
    ex, ey, ez = sp.identity(nx, format='csr'), sp.identity(ny, format='csr'), sp.identity(nz, format='csr')
    dx = sp.kron(ez, sp.kron(ey, d1(nx, cx)), format='csr')
    dy = sp.kron(ez, sp.kron(d1(ny, cy), ex), format='csr')
    ks = sp.csr_matrix(-(dx - ratio * dy))
    ks.sort_indices()
    return ks


def buckling_load_factors(nx, ny, nz, ax, ay, az, ratio=0.3):
    """All load factors alpha of (lap3d + alpha stress_stiffness) v = 0, positive ones ascending first: mode (i, j, k) has
    alpha = (lx_i + ly_j + lz_k) / (lx_i - ratio ly_j), l the eigenvalues of the 1-D second differences."""
    cx, cy, cz = lap3d_coefficients(nx, ny, nz, ax, ay, az)
    lx = 2 * cx * (1 - np.cos(np.arange(1, nx + 1) * np.pi / (nx + 1)))
    ly = 2 * cy * (1 - np.cos(np.arange(1, ny + 1) * np.pi / (ny + 1)))
    lz = 2 * cz * (1 - np.cos(np.arange(1, nz + 1) * np.pi / (nz + 1)))
    num = lx[:, None, None] + ly[None, :, None] + lz[None, None, :]
    den = (lx[:, None, None] - ratio * ly[None, :, None]) + 0 * lz[None, None, :]
    alpha = (num / den).ravel()
    return np.sort(alpha[alpha > 0]), np.sort(alpha[alpha < 0])[::-1]


def _hash01(i, j):
    """Deterministic pseudo-random numbers in [0, 1) from two int64 arrays (splitmix64 finaliser)."""
    z = (i.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ (j.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


FE_SURROGATE_GRID = (23, 46, 85)     # nodes; 2 unknowns per node => n = 179 860 = shipsec5's order


def fe_surrogate(grid=FE_SURROGATE_GRID, dof=2, dtype=np.float64):
    """FE-like symmetric positive definite surrogate for SuiteSparse shipsec5 (BASELINE config 3).

    `dof` unknowns per node of a 3-D structured grid of nodes, unknowns of a node numbered
    consecutively, nodes in lexicographic (x fastest) order -- the banded structure a
    bandwidth-reducing ordering gives a ship-section FE model.  A node is coupled to its 26 nearest
    neighbours and to the second neighbours along x (29 nodes x dof = 58 entries per interior row;
    ~55.0 per row on average for the default grid: n = 179 860, nnz ~ 9.9 M, where shipsec5 has
    n = 179 860, ~56 per row).  Off-diagonal entries are negative with hashed weights (symmetric by
    construction), the diagonal is the absolute row sum of the FULL stencil (couplings that leave
    the grid keep their share: a clamped boundary) plus 1 %, so the matrix is strictly diagonally
    dominant, hence positive definite, with a spectrum that starts well above zero like a
    constrained stiffness matrix."""
    gx, gy, gz = grid
    nn = gx * gy * gz
    n = nn * dof
    node = np.arange(nn, dtype=np.int64)
    ix, iy, iz = node % gx, (node // gx) % gy, node // (gx * gy)
    offsets = [(dx, dy, dz) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)] + [(-2, 0, 0), (2, 0, 0)]
    rows, cols, vals = [], [], []
    diag = np.zeros(n, dtype=np.float64)
    for dx, dy, dz in offsets:
        base = 1.0 / (1.0 + dx * dx + dy * dy + dz * dz)        # decays with the node distance
        inside = (ix + dx >= 0) & (ix + dx < gx) & (iy + dy >= 0) & (iy + dy < gy) & (iz + dz >= 0) & (iz + dz < gz)
        a = node[inside]
        b = a + dx + dy * gx + dz * gx * gy
        for da in range(dof):
            for db in range(dof):
                if dx == 0 and dy == 0 and dz == 0 and da == db:
                    continue                                     # the diagonal is set below
                # the clamped share of couplings that leave the grid
                diag[node * dof + da] += np.where(inside, 0.0, base * 1.25)
                ra, cb = a * dof + da, b * dof + db
                w = base * (0.75 + 0.5 * _hash01(np.minimum(ra, cb), np.maximum(ra, cb)))
                rows.append(ra); cols.append(cb); vals.append(-w)
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    diag += np.bincount(rows, weights=-vals, minlength=n)
    diag *= 1.01
    A = sp.csr_matrix((np.concatenate((vals, diag)), (np.concatenate((rows, np.arange(n))),
                                                        np.concatenate((cols, np.arange(n))))), shape=(n, n))
    A.sort_indices()
    return A.astype(dtype)


def read_matrix_market(path):
    """Coordinate-format Matrix-Market file (real / complex / integer / pattern; general /
    symmetric / hermitian / skew-symmetric) -> scipy CSR with both triangles, so that a supplied
    SuiteSparse file (e.g. shipsec5.mtx, BASELINE config 3) drops into SparseSymmetricMatrix.
    Own reader (the reference reads its matrices with scipy.io, raleigh/examples/eigenproblems):
    one pass with numpy, ~10 M entries in a few seconds; .gz files are read transparently."""
    opener = gzip.open if str(path).endswith('.gz') else open
    with opener(path, 'rt') as fh:
        header = fh.readline().split()
        if len(header) < 5 or header[0] != '%%MatrixMarket' or header[1].lower() != 'matrix':
            raise ValueError('%s: not a Matrix-Market matrix file' % path)
        fmt, field, symmetry = header[2].lower(), header[3].lower(), header[4].lower()
        if fmt != 'coordinate':
            raise ValueError('%s: only the coordinate (sparse) format is supported' % path)
        if field not in ('real', 'double', 'complex', 'integer', 'pattern'):
            raise ValueError('%s: unknown field %s' % (path, field))
        if symmetry not in ('general', 'symmetric', 'hermitian', 'skew-symmetric'):
            raise ValueError('%s: unknown symmetry %s' % (path, symmetry))
        line = fh.readline()
        while line.startswith('%') or not line.strip():
            line = fh.readline()
        nr, nc, nz = (int(t) for t in line.split()[:3])
        ncol = {'pattern': 2, 'complex': 4}.get(field, 3)
        data = np.loadtxt(fh, dtype=np.float64, ndmin=2, comments='%') if nz > 0 else np.zeros((0, ncol))
    if data.shape[0] != nz or (nz > 0 and data.shape[1] < ncol):
        raise ValueError('%s: expected %d entries of %d columns, found %s' % (path, nz, ncol, data.shape))
    i, j = data[:, 0].astype(np.int64) - 1, data[:, 1].astype(np.int64) - 1
    if nz and (i.min() < 0 or j.min() < 0 or i.max() >= nr or j.max() >= nc):
        raise ValueError('%s: index out of range' % path)
    if field == 'pattern':
        v = np.ones(nz)
    elif field == 'complex':
        v = data[:, 2] + 1j * data[:, 3]
    else:
        v = data[:, 2]
    if symmetry != 'general':
        off = i != j
        mirror = {'symmetric': v[off], 'hermitian': np.conj(v[off]), 'skew-symmetric': -v[off]}[symmetry]
        i, j, v = np.concatenate((i, j[off])), np.concatenate((j, i[off])), np.concatenate((v, mirror))
    a = sp.csr_matrix((v, (i, j)), shape=(nr, nc))      # duplicate entries are summed, as the format specifies
    a.sort_indices()
    return a
