"""Host image of the column-blocked tile format (spectra_amd/csrc/tiles.hip) through the C ABI — no device needed.

The builder is integer work (bucketing, packing, padding): checked exactly.  The host product walks the tiles in the
kernel's order; a row's products are added in ascending column order = CSR storage order, so it must equal the oracle's CSR
row-dot BIT FOR BIT."""
import numpy as np
import scipy.sparse as sp

import oracle as O
import spectra_amd as sa


def scattered(n, per_row, seed, ncols=None):
    ncols = ncols or n
    rng = np.random.default_rng(seed)
    r = np.repeat(np.arange(n), per_row)
    c = rng.integers(0, ncols, r.size)
    A = sp.coo_matrix((rng.uniform(-1, 1, r.size), (r, c)), shape=(n, ncols)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A


def test_tiles_product_is_the_csr_row_dot_bit_for_bit():
    for n, ncols, per_row, seed in [(5000, 3_000_000, 9, 0), (300_000, 300_000, 7, 1), (4096 * 3 + 17, 5_000_000, 15, 2)]:
        A = scattered(n, per_row, seed, ncols)
        x = np.random.default_rng(seed + 10).standard_normal(ncols)
        y, st = sa.tiles_spmv_host(A, x)
        assert y is not None and st["entries"] - st["padding"] == A.nnz
        ref = O.Op.csr(n, ncols, A.indptr, A.indices, A.data).perform_op(x)
        assert np.array_equal(y, ref)
        assert st["padding"] <= 0.05 * A.nnz + 64 * st["chunks"]


def test_tiles_runs_clusters_empty_rows_and_rectangular_shapes():
    # rows with several entries inside one column block (runs up to 7), empty rows, a wide rectangular matrix
    n, ncols = 10_000, 700_000
    rng = np.random.default_rng(5)
    rows, cols = [], []
    for r in range(0, n, 3):                       # every third row: 3 clusters of 1..7 neighbours
        for blk in rng.choice(5, 3, replace=False):        # three different column blocks
            base = int(blk) * 131072 + int(rng.integers(0, 131072 - 10))
            k = int(rng.integers(1, 8))
            rows += [r] * k
            cols += list(range(base, base + k))
    A = sp.coo_matrix((rng.uniform(-1, 1, len(rows)), (rows, cols)), shape=(n, ncols)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    x = rng.standard_normal(ncols)
    y, st = sa.tiles_spmv_host(A, x)
    assert y is not None
    assert np.array_equal(y, O.Op.csr(n, ncols, A.indptr, A.indices, A.data).perform_op(x))
    assert np.all(y[1::3] == 0.0)


def test_tiles_long_runs_and_veto():
    # rows with more than 7 entries inside one column block are emitted in several passes (3-bit run length per head)
    rng = np.random.default_rng(9)
    n, ncols = 300, 400_000
    rows, cols = [], []
    for r in range(n):
        k = int(rng.integers(1, 40))
        cs = np.sort(rng.choice(2000, k, replace=False)) + 131072 * int(rng.integers(0, 3))
        rows += [r] * k
        cols += cs.tolist()
    A = sp.coo_matrix((rng.uniform(-1, 1, len(rows)), (rows, cols)), shape=(n, ncols)).tocsr()
    A.sort_indices()
    x = rng.standard_normal(ncols)
    y, st = sa.tiles_spmv_host(A, x)
    assert y is not None and st["entries"] - st["padding"] == A.nnz
    assert np.array_equal(y, O.Op.csr(n, ncols, A.indptr, A.indices, A.data).perform_op(x))
    # unsorted rows: the run order would not be the CSR storage order -> not built
    B = sp.csr_matrix((np.ones(3), np.array([5, 2, 9]), np.array([0, 3])), shape=(1, 10))
    yb, _ = sa.tiles_spmv_host(B, np.ones(10))
    assert yb is None
