"""Synthetic host-side matrices of the secondary workloads (bench.py, tools/, tests): none of them has the fixed-offset structure
of the headline matrix M-band, so none of them can use its diagonal storage.

* m_rand:       every row has 7 partners at uniformly random columns, symmetrised — all far gathers (SURVEY.md 8d "M-rand")
* jitter_band:  M-band's 14 off-diagonals with every entry moved by a per-entry jitter of at most +-64 columns, symmetrised:
                variable row lengths, ~1800 distinct diagonals (no offset codes, no diagonal storage), columns local but not on
                constant offsets — what a mesh matrix looks like after a bandwidth-reducing ordering
* stencil7:     3-D 7-point stencil on an m^3 grid with random symmetric values, natural order
"""
import numpy as np
import scipy.sparse as sp

BAND_OFFSETS = (1, 2, 3, 1000, 1001, 100000, 100001)


def m_rand(n, seed=20240607):
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n, dtype=np.int64), 7)
    cols = rng.integers(0, n, size=rows.size, dtype=np.int64)
    vals = rng.uniform(-0.5, 0.5, size=rows.size)
    U = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    U.sum_duplicates()
    A = (U + U.T + sp.diags(rng.uniform(-0.5, 0.5, n))).tocsr()
    A.sort_indices()
    return A


def jitter_band(n, seed=20250925, jitter=64, offsets=BAND_OFFSETS):
    rng = np.random.default_rng(seed)
    i = np.arange(n, dtype=np.int64)
    rows, cols = [], []
    for off in offsets:
        c = i + off + rng.integers(-jitter, jitter + 1, size=n, dtype=np.int64)
        keep = (c >= 0) & (c < n) & (c != i)
        rows.append(i[keep])
        cols.append(c[keep])
    rows = np.concatenate(rows)
    cols = np.concatenate(cols)
    vals = rng.uniform(-0.5, 0.5, size=rows.size)
    U = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    U.sum_duplicates()
    A = (U + U.T + sp.diags(rng.uniform(-0.5, 0.5, n))).tocsr()
    A.sort_indices()
    return A


def stencil7(m, seed=0):
    I = sp.identity(m, format="csr")
    T = sp.diags([np.ones(m - 1), np.ones(m - 1)], [-1, 1], format="csr")
    A = (sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T) + 6 * sp.identity(m ** 3)).tocsr()
    rng = np.random.default_rng(seed)
    A.data[:] = rng.uniform(-0.5, 0.5, A.nnz)
    A = (sp.tril(A) + sp.tril(A, -1).T).tocsr()
    A.sort_indices()
    return A
