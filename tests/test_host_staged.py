"""Host image of the staged SpMV format (spectra_amd/csrc/staged.hip: two streaming phases, x and y in LDS) through the C ABI — no
device needed.

The builder is integer work (bucketing by column block, batches of <= 1024 entries per row bin, ranks of the entries of one row
inside a batch): the host product walks the image in the order of the two kernels — products column block by column block, then
per row bin batch after batch and rank after rank — which adds a row's products in ascending column order with one accumulator
= the CSR storage order, so it must equal the oracle's CSR row-dot BIT FOR BIT."""
import numpy as np
import scipy.sparse as sp

import oracle as O
import spectra_amd as sa
from test_host_tiles import scattered


def check(A, seed=0):
    n, ncols = A.shape
    x = np.random.default_rng(seed + 10).standard_normal(ncols)
    y, st = sa.staged_spmv_host(A, x)
    assert y is not None
    assert np.array_equal(y, O.Op.csr(n, ncols, A.indptr, A.indices, A.data).perform_op(x))
    # bins of 8192 rows, lower ones (a multiple of 256 rows) when there would be fewer than 512 of them
    height = 8192 if (n + 8191) // 8192 >= 512 else min(8192, max(256, -(-(-(-n // 512)) // 256) * 256))
    assert st["bins"] == -(-n // height) and A.nnz <= st["slots"] <= A.nnz + (ncols + 8191) // 8192
    return st


def test_staged_product_is_the_csr_row_dot_bit_for_bit():
    for n, ncols, per_row, seed in [(5000, 3_000_000, 9, 0), (300_000, 300_000, 7, 1), (8192 * 3 + 17, 5_000_000, 15, 2)]:
        st = check(scattered(n, per_row, seed, ncols), seed)
        # scattered columns: a batch collects the bin's pieces of several column blocks, few entries of a row meet in one batch
        assert st["max_rounds"] <= 8 and st["batches"] >= st["bins"]


def test_staged_clusters_dense_rows_empty_rows_and_rectangular_shapes():
    rng = np.random.default_rng(5)
    # (1) every third row holds clusters of neighbours inside one column block: ranks up to the cluster length, then new batches
    n, ncols = 20_000, 700_000
    rows, cols = [], []
    for r in range(0, n, 3):
        for blk in rng.choice(80, 3, replace=False):
            base = int(blk) * 8192 + int(rng.integers(0, 8192 - 40))
            k = int(rng.integers(1, 30))  # more than 8 entries of one row in one block: the rank field overflows -> batch split
            rows += [r] * k
            cols += list(range(base, base + k))
    A = sp.coo_matrix((rng.uniform(-1, 1, len(rows)), (rows, cols)), shape=(n, ncols)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    st = check(A, 1)
    assert st["max_rounds"] == 8
    y, _ = sa.staged_spmv_host(A, np.ones(ncols))
    assert np.all(y[1::3] == 0.0)
    # (2) a few dense rows (every column) among sparse ones, one row bin: batches of exactly 1024 entries, rank restarts
    n, ncols = 600, 50_000
    D = sp.lil_matrix((n, ncols))
    for r in (0, 17, 599):
        D[r, :] = rng.uniform(-1, 1, ncols)
    S = scattered(n, 5, 3, ncols).tolil()
    for r in (0, 17, 599):
        S[r, :] = 0
    A = (D + S).tocsr()
    A.sort_indices()
    A.eliminate_zeros()
    st = check(A, 2)
    assert st["max_rounds"] == 8
    # (3) banded matrix (every entry of a row in one or two column blocks), a single column, an empty matrix row range
    n = 30_000
    B = sp.diags([rng.uniform(-1, 1, n - abs(o)) for o in (-100, -1, 0, 1, 100)], [-100, -1, 0, 1, 100]).tocsr()
    B.sort_indices()
    check(B, 3)
    C1 = sp.csr_matrix((rng.uniform(-1, 1, 1000), (np.arange(0, 20_000, 20), np.zeros(1000, dtype=int))), shape=(20_000, 1))
    check(C1, 4)


def test_staged_veto_on_unsorted_rows():
    A = scattered(2000, 6, 7, 100_000)
    A.indices[A.indptr[5]:A.indptr[5] + 2] = A.indices[A.indptr[5]:A.indptr[5] + 2][::-1].copy()
    A.has_sorted_indices = False
    y, _ = sa.staged_spmv_host(A, np.ones(100_000))
    assert y is None  # the row sums would not follow the storage order: the format declines, the CSR kernels take the matrix


def test_heavy_rows_among_scattered_ones_are_flagged_for_the_automatic_choice():
    # a row with more than 8 entries inside one batch closes the batch; a few dense rows among scattered ones give one
    # nearly empty batch per 8 of their entries.  The image stays correct (and is checked), but is reported as poorly filled, and
    # the automatic format choice at ingest (csr.hip upload_rows) declines it.
    n = 200_000
    S = scattered(n, 8, 4, n)
    base = sa.staged_spmv_host(S, np.ones(n))[1]
    assert base["well_filled"]
    rng = np.random.default_rng(1)
    dense_rows = np.array([5, 70_000, 130_001, n - 1])
    D = sp.coo_matrix((rng.uniform(-1, 1, 4 * n), (np.repeat(dense_rows, n), np.tile(np.arange(n), 4))), shape=(n, n)).tocsr()
    keep = np.ones(n, bool)
    keep[dense_rows] = False
    A = (sp.diags(keep.astype(float)) @ S + D).tocsr()
    A.sort_indices()
    st = check(A, 3)
    assert not st["well_filled"] and st["batches"] > 20 * base["batches"]
