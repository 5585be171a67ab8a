"""The int32 CSR kernel with the x entries of a row-block staged through LDS windows (csr.hip k_spmv_csr_win):
the product of SparseSymMatProd / SparseGenMatProd::perform_op (MatOp/SparseSymMatProd.h:83-88, SparseGenMatProd.h:72-77) on the
plain int32 CSR arrays.  Every case is compared BIT FOR BIT with the oracle's CSR row-dot, with the windows on and off (the gather
kernel k_spmv_csr_stream), and the tests assert which kernel ran through `windows_info()`.
"""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle as O
import spectra_amd as sa
from spectra_amd import workloads
from helpers import check_window_records

pytestmark = [pytest.mark.gpu, pytest.mark.operator_only]


def oracle_product(A, x):
    A = A.tocsr()
    A.sort_indices()
    return O.Op.csr(A.shape[0], A.shape[1], A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data).perform_op(x)


def device_product(op, x):
    import torch

    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    y = torch.full((op.local_rows() + 2,), np.nan, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    op.spmv_device(xd.data_ptr(), y.data_ptr())
    op.ctx.sync()
    return y[: op.local_rows()].cpu().numpy()


def both_kernels(op, A, seeds=(0, 1), expect_windows=True):
    info = op.windows_info()
    assert (info["lds_doubles"] > 0) == expect_windows, info
    if expect_windows and A.shape[0] <= 700000 and op.reordering() == "none":
        check_table(op, A)
    op.set_spmv_format(0)
    try:
        for seed in seeds:
            rng = np.random.default_rng(seed)
            x = rng.standard_normal(A.shape[1]) * np.exp(rng.uniform(-8, 8, A.shape[1]))
            ref = oracle_product(A, x)
            for windows in (True, False):
                op.use_windows(windows)
                y = device_product(op, x)
                assert np.array_equal(y, ref), (windows, int(np.count_nonzero(y != ref)), np.nanmax(np.abs(y - ref)))
    finally:
        op.use_windows(None)
        op.set_spmv_format(-1)
    return info


def check_table(op, S):
    """The device's table: the records' invariants against the stored matrix S, and equality with the table the HOST builds from
    the same arrays with the same selection code (mispec_csr_windows_host)."""
    T = op.windows_table()
    covered = check_window_records(T, S)
    assert covered == op.windows_info()["covered_entries"]
    Sc = S.tocsr()
    Sc.sort_indices()
    assert np.array_equal(T, sa.windows_host(Sc))


def local_random(n, per_row, spread, seed, far=0):
    """Rows with about `per_row` entries at columns within +-spread of the diagonal (ragged: 0 .. 2 per_row), plus `far`
    entries per row anywhere in the matrix."""
    rng = np.random.default_rng(seed)
    counts = rng.integers(0, 2 * per_row + 1, n)
    rows = np.repeat(np.arange(n), counts)
    cols = np.clip(rows + rng.integers(-spread, spread + 1, rows.size), 0, n - 1)
    if far:
        fr = np.repeat(np.arange(n), far)
        rows = np.concatenate([rows, fr])
        cols = np.concatenate([cols, rng.integers(0, n, fr.size)])
    A = sp.coo_matrix((rng.uniform(-1, 1, rows.size), (rows, cols)), shape=(n, n)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A


@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 1000, 4097, 70001])
def test_ragged_local_patterns(ctx, n):
    A = local_random(n, 7, 300, n)
    op = sa.SparseGenMatProd(A, ctx=ctx)
    both_kernels(op, A, expect_windows=n >= 255)  # a handful of entries is not worth a window


def test_automatic_rule_short_rows_keep_the_gather_kernel(ctx):
    # rows of fewer than 9 entries on average do not pay for their windows (7-point stencil after RCM: 0.198-0.203 ms with windows,
    # 0.194 ms with gathers in the solver loop): the table is adopted, the automatic choice keeps k_spmv_csr_stream, the switch
    # still selects either kernel, and all three settings give the same bits
    n = 70001
    short = local_random(n, 3, 300, 5)   # ~3 entries per row
    longer = local_random(n, 7, 300, 6)  # ~7: 0 .. 14 entries, 7 on average -> below 9 as well
    dense = local_random(n, 12, 300, 7)  # ~12
    for A, auto in ((short, False), (longer, False), (dense, True)):
        op = sa.SparseGenMatProd(A, ctx=ctx)
        assert op.windows_info()["lds_doubles"] > 0 and op.windows_in_use() == auto
        op.use_windows(True)
        assert op.windows_in_use()
        op.use_windows(False)
        assert not op.windows_in_use()
        op.use_windows(None)
        assert op.windows_in_use() == auto
        both_kernels(op, A, seeds=(0,))
        x = np.random.default_rng(3).standard_normal(n)
        op.set_spmv_format(0)
        assert np.array_equal(device_product(op, x), oracle_product(A, x))
        op.set_spmv_format(-1)


def test_odd_column_count_and_windows_at_both_ends(ctx):
    # windows hold pairs of doubles: the last pair of an odd-length x is clipped; columns 0 and n - 1 are referenced by every block
    n = 3333
    A = local_random(n, 5, 40, 3).tolil()
    A[:, 0] = 0.5
    A[:, n - 1] = -0.25
    A = A.tocsr()
    op = sa.SparseGenMatProd(A, ctx=ctx)
    both_kernels(op, A)


def test_rectangular(ctx):
    rng = np.random.default_rng(5)
    A = sp.random(1500, 911, density=0.01, random_state=5, format="csr")
    A.data[:] = rng.uniform(-1, 1, A.nnz)
    op = sa.SparseGenMatProd(A, ctx=ctx)
    both_kernels(op, A)


def test_rows_longer_than_a_chunk_and_many_windows(ctx):
    # one row of 5000 entries, another block whose columns form far more than 8 clusters (merged down to 8 windows)
    n = 200000
    rng = np.random.default_rng(11)
    rows = [np.full(5000, 700), np.repeat(np.arange(1024, 1280), 40), np.arange(n)]
    cols = [np.sort(rng.choice(np.arange(400, 9000), 5000, replace=False)),
            np.tile(np.arange(40) * 1000, 256) + np.repeat(np.arange(1024, 1280), 40), np.arange(n)]
    r, c = np.concatenate(rows), np.concatenate(cols)
    A = sp.coo_matrix((rng.uniform(-1, 1, r.size), (r, c)), shape=(n, n)).tocsr()
    A.sum_duplicates()
    op = sa.SparseGenMatProd(A, ctx=ctx)
    both_kernels(op, A)


def test_mixed_blocks_far_entries_keep_the_gather(ctx):
    # every row has local entries and two entries anywhere in a 600000-column matrix: blocks are flagged "far", their local entries
    # come from LDS, the rest from global memory; 6 of 7 entries are covered, so the windows are adopted
    n = 600000
    A = local_random(n, 6, 500, 21, far=1)
    op = sa.SparseGenMatProd(A, ctx=ctx)
    info = both_kernels(op, A, seeds=(0,))
    assert 0.5 * A.nnz < info["covered_entries"] < A.nnz


def test_scattered_pattern_declines_the_windows(ctx):
    n = 600000
    A = workloads.m_rand(n, seed=3)
    op = sa.SparseGenMatProd(A, ctx=ctx, reorder="none")
    both_kernels(op, A, seeds=(0,), expect_windows=False)


def test_blocks_whose_windows_exceed_the_lds_budget_fall_back(ctx):
    # 256 rows whose columns cover 40000 contiguous columns (more than 6144 doubles of windows): that block gathers, the rest
    # of the matrix keeps its windows
    n = 100000
    rng = np.random.default_rng(9)
    wide_r = np.repeat(np.arange(5120, 5376), 160)
    wide_c = (np.tile(np.arange(160) * 250, 256) + rng.integers(0, 250, wide_r.size)) + 5000
    band = sp.diags([rng.uniform(-1, 1, n - abs(k)) for k in (-3, -1, 0, 1, 3)], [-3, -1, 0, 1, 3], format="csr")
    A = (band + sp.coo_matrix((rng.uniform(-1, 1, wide_r.size), (wide_r, wide_c)), shape=(n, n))).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    op = sa.SparseGenMatProd(A, ctx=ctx)
    info = both_kernels(op, A, seeds=(0,))
    assert info["blocks"] == (n + 255) // 256 - 1


def test_the_solver_loop_is_identical_with_and_without_windows(ctx):
    # the fused Lanczos epilogue (w -= beta v_prev, partial <v, w>) of both kernels: same records, same solve bit for bit
    n = 300000
    A = workloads.jitter_band(n, offsets=(1, 2, 3, 1000, 1001, 20000, 20001))
    op = sa.SparseSymMatProd(sp.tril(A).tocsc(), ctx=ctx)
    assert op.spmv_format() == 0 and op.windows_info()["lds_doubles"] > 0
    out = []
    for windows in (True, False):
        op.use_windows(windows)
        e = sa.SymEigsSolver(op, 6, 20)
        e.init()
        nconv = e.compute(sa.SortRule.LargestMagn, 300, 1e-10)
        assert nconv == 6
        out.append((e.eigenvalues(), e.num_operations(), e.eigenvectors()))
        assert e.residuals().max() <= 1e-9
    op.use_windows(None)
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1] and np.array_equal(out[0][2], out[1][2])


@pytest.mark.parametrize("which", ["jitter_band", "stencil_rcm"])
def test_full_size_irregular_local_matrices_bit_exact(ctx, which):
    # at n = 1e7 the jittered band (no fixed offsets: format 0 is the automatic choice) and the 7-point stencil
    # in random order, reordered at ingest (reverse Cuthill-McKee), against the oracle's row-dot
    if which == "jitter_band":
        A = workloads.jitter_band(10_000_000)
        op = sa.SparseSymMatProd(sp.tril(A).tocsc(), ctx=ctx)
        assert op.reordering() == "none"
    else:
        B = workloads.stencil7(215)
        perm = np.random.default_rng(1).permutation(B.shape[0])
        A = B[perm][:, perm].tocsr()
        A.sort_indices()
        del B
        op = sa.SparseSymMatProd(sp.tril(A).tocsc(), ctx=ctx)
        assert op.reordering() == "rcm"
    assert op.spmv_format() == 0
    info = op.windows_info()
    assert info["covered_entries"] >= 0.999 * op.nnz() and info["lds_doubles"] > 0  # (thin level sets at the ends of the RCM order are gathered)
    n = A.shape[0]
    x = O.simple_random(n, 0)
    if which == "stencil_rcm":
        # the stored matrix is P A P' with its rows sorted by the NEW column index: that is the summation order the library
        # guarantees for a reordered matrix (tests/test_gpu_reorder.py), so the oracle runs on the stored matrix
        p = op.permutation()
        S = A[p][:, p].tocsr()
        ref = np.empty(n)
        ref[p] = oracle_product(S, x[p])
        check_table(op, S)
    else:
        ref = oracle_product(A, x)
        check_table(op, A)
    for windows in (True, False):
        op.use_windows(windows)
        assert np.array_equal(op.perform_op(x), ref), windows
    op.use_windows(None)
