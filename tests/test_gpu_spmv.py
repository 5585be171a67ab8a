"""K1 — the CSR SpMV kernel behind Sparse{Sym,Gen}MatProd::perform_op, through the C ABI, vs the CPU oracle.

The kernel adds each row's products in storage order, exactly like the oracle's row-dot (and Eigen's
row-major product), so agreement with oracle.Op.csr is BIT-EXACT.  Against the reference-semantics
CSC-lower self-adjoint product (different summation order) the bound is 1e-13 relative.
"""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle as O
import spectra_amd as sa
from helpers import SPARSE_CASES, sparse_fixture

pytestmark = pytest.mark.gpu


def rand_x(n, seed=0):
    return np.random.default_rng(seed).uniform(-1, 1, n)


@pytest.mark.parametrize("n,prob", [(c[0], c[1]) for c in SPARSE_CASES])
def test_sym_op_on_reference_fixtures(ctx, n, prob):
    # test/SparseSymMatProd.cpp:37-54 (op * M == mat * M, op(i,j) == coeff) on the reproducible fixture;
    # only the lower triangle of the (non-symmetric) input may be read (test/SymEigs.cpp:27-28)
    A, S = sparse_fixture(n, prob)
    op = sa.SparseSymMatProd(A, ctx=ctx)
    assert (op.rows(), op.cols()) == (n, n)
    x = rand_x(n)
    y = op.perform_op(x)
    ref_sym = O.Op.csc_sym(n, A.indptr, A.indices, A.data, True).perform_op(x)
    assert np.abs(y - ref_sym).max() <= 1e-13 * max(1.0, np.abs(ref_sym).max())
    ref_csr = O.Op.csr(n, n, S.indptr, S.indices, S.data).perform_op(x)
    assert np.array_equal(y, ref_csr)  # bit-exact: same summation order
    X = np.random.default_rng(1).uniform(-1, 1, (n, 3))
    assert np.allclose(op @ X, S @ X, rtol=0, atol=1e-13)
    i, j = min(45, n - 1), min(22, n - 2)
    assert op(i, j) == S[i, j] and op(j, i) == S[j, i]
    # upper-triangle operator of the same input
    Su = (sp.triu(A) + sp.triu(A, 1).T).tocsr()
    opu = sa.SparseSymMatProd(A, uplo="U", ctx=ctx)
    assert np.array_equal(opu.perform_op(x), O.Op.csr(n, n, Su.indptr, Su.indices, Su.data).perform_op(x))
    # row-major input (Flags = RowMajor)
    opr = sa.SparseSymMatProd(A.tocsr(), ctx=ctx)
    assert np.array_equal(opr.perform_op(x), ref_csr)


@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_gen_op(ctx, fmt):
    # test/SparseGenMatProd.cpp:37-53
    n = 300
    A = sp.random(n, n, density=0.05, random_state=3, format=fmt)
    A.data[:] = np.random.default_rng(3).uniform(-1, 1, A.nnz)
    op = sa.SparseGenMatProd(A, ctx=ctx)
    x = rand_x(n, 4)
    Ar = A.tocsr()
    Ar.sort_indices()
    assert np.array_equal(op.perform_op(x), O.Op.csr(n, n, Ar.indptr, Ar.indices, Ar.data).perform_op(x))
    Ac = A.tocsc()
    Ac.sort_indices()
    ref_csc = O.Op.csc(n, n, Ac.indptr, Ac.indices, Ac.data).perform_op(x)  # the reference default layout
    assert np.abs(op.perform_op(x) - ref_csc).max() <= 1e-13


@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 511, 1000, 4097])
def test_ragged_sizes_and_empty_rows(ctx, n):
    rng = np.random.default_rng(n)
    A = sp.random(n, n, density=min(1.0, 8.0 / n), random_state=n, format="csr")
    A.data[:] = rng.uniform(-1, 1, A.nnz)
    if n > 4:  # force some empty rows
        A = A.tolil()
        A[n // 2, :] = 0
        A[0, :] = 0
        A = A.tocsr()
        A.eliminate_zeros()
    op = sa.SparseGenMatProd(A, ctx=ctx)
    x = rand_x(n, 9)
    A.sort_indices()
    assert np.array_equal(op.perform_op(x), O.Op.csr(n, n, A.indptr, A.indices, A.data).perform_op(x))


def test_rows_longer_than_the_lds_chunk(ctx):
    # a dense-ish row (> 4080 products) must be summed across chunks, still in storage order
    n = 6000
    rng = np.random.default_rng(7)
    rows = [np.full(n, 3), np.full(5000, 700), rng.integers(0, n, 20000)]
    cols = [np.arange(n), rng.choice(n, 5000, replace=False), rng.integers(0, n, 20000)]
    r, c = np.concatenate(rows), np.concatenate(cols)
    A = sp.coo_matrix((rng.uniform(-1, 1, len(r)), (r, c)), shape=(n, n)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    op = sa.SparseGenMatProd(A, ctx=ctx)
    x = rand_x(n, 11)
    assert np.array_equal(op.perform_op(x), O.Op.csr(n, n, A.indptr, A.indices, A.data).perform_op(x))


def test_synthetic_band_matrix_is_bit_identical_to_the_cpu_statement(ctx):
    n = 300000
    op = sa.SparseSymMatProd.synth_band(n, ctx=ctx)
    rp, ci, v = op.to_host_csr()
    rp0, ci0, v0 = O.synth_band_csr(n)
    assert np.array_equal(rp, rp0) and np.array_equal(ci, ci0) and np.array_equal(v, v0)
    x = rand_x(n, 2)
    assert np.array_equal(op.perform_op(x), O.Op.csr(n, n, rp0, ci0, v0).perform_op(x))
    g = sa.SparseGenMatProd.synth_band(5000, offsets=(1, 2, 50), ctx=ctx)
    rp1, ci1, v1 = g.to_host_csr()
    rp2, ci2, v2 = O.synth_band_csr(5000, offsets=(1, 2, 50), symmetric=False)
    assert np.array_equal(ci1, ci2) and np.array_equal(v1, v2)


def test_spmv_linearity_at_full_size(ctx):
    # size-independent property at BASELINE.json's n = 1e7 (the oracle cannot be run there in seconds):
    # A(a x + b y) == a A x + b A y up to rounding, and symmetry x'Ay == y'Ax.
    import torch

    n = 10_000_000
    op = sa.SparseSymMatProd.synth_band(n, ctx=ctx)
    assert op.nnz() == 15 * n - 2 * (1 + 2 + 3 + 1000 + 1001 + 100000 + 100001)
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) - 0.5
    y = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) - 0.5
    Ax, Ay, Az = (torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(3))
    torch.cuda.synchronize()
    z = 0.75 * x - 1.25 * y
    torch.cuda.synchronize()
    op.spmv_device(x.data_ptr(), Ax.data_ptr())
    op.spmv_device(y.data_ptr(), Ay.data_ptr())
    op.spmv_device(z.data_ptr(), Az.data_ptr())
    ctx.sync()
    lin = (Az - (0.75 * Ax - 1.25 * Ay)).abs().max().item()
    assert lin < 1e-14 * 16
    sym = abs(torch.dot(x, Ay).item() - torch.dot(y, Ax).item())
    assert sym < 1e-9 * abs(torch.dot(x, Ay).item()) + 1e-7
    # spot-check 1000 rows against the CPU statement of the matrix
    rows = np.random.default_rng(0).integers(0, n, 1000)
    xh = x.cpu().numpy()
    offs = np.array(sorted({0} | {s * o for o in sa.BAND_OFFSETS for s in (1, -1)}))
    Axh = Ax.cpu().numpy()
    for i in rows:
        acc = 0.0
        for o in offs:
            j = i + o
            if 0 <= j < n:
                acc += O.lib().oracle_synth_value(sa.SYNTH_SEED, min(i, j), max(i, j)) * xh[j]
        assert Axh[i] == acc


# ---- offset-coded index format (one byte per entry, col = row + dict[code]) -------------------------------------
def _both_formats(op, x):
    """The product in every storage format the matrix has: returns (offset-coded, int32); the diagonal format (when it
    was built, it is the automatic choice) is asserted equal to the offset-coded one on the way."""
    assert op.offset_codes() > 0
    auto = op.spmv_format()
    y_auto = op.perform_op(x)
    op.set_spmv_format(1)
    assert op.spmv_format() == 1
    y_codes = op.perform_op(x)
    op.set_spmv_format(0)
    assert op.spmv_format() == 0 and op.offset_codes() == 0
    y_plain = op.perform_op(x)
    op.set_spmv_format(-1)
    assert op.spmv_format() == auto
    assert np.array_equal(y_auto, y_codes)
    return y_codes, y_plain


def test_offset_codes_are_chosen_for_diagonal_structure_only(ctx):
    n = 5000
    band = sa.SparseSymMatProd.synth_band(n, offsets=(1, 2, 50), ctx=ctx)
    assert band.offset_codes() == 7 and band.stored_bytes() < band.algorithmic_bytes()
    assert band.spmv_format() == 2  # few, full diagonals: diagonal storage
    A, _ = sparse_fixture(1000, 0.01)  # ~ 2000 distinct diagonals: stays on int32 indices
    op = sa.SparseSymMatProd(A, ctx=ctx)
    assert op.offset_codes() == 0 and op.stored_bytes() == op.algorithmic_bytes() and op.spmv_format() == 0


@pytest.mark.parametrize("n", [3, 255, 256, 257, 1000, 4097, 300000])
def test_offset_coded_kernel_is_bit_identical_on_band_matrices(ctx, n):
    offs = tuple(o for o in (1, 2, 3, 1000, 1001, 100000, 100001) if o < n)
    op = sa.SparseSymMatProd.synth_band(n, offsets=offs, ctx=ctx)
    x = rand_x(n, 3)
    y_codes, y_plain = _both_formats(op, x)
    assert np.array_equal(y_codes, y_plain)
    if n <= 5000:
        rp, ci, v = O.synth_band_csr(n, offsets=offs)
        assert np.array_equal(y_codes, O.Op.csr(n, n, rp, ci, v).perform_op(x))


def test_offset_codes_from_host_uploads(ctx):
    # banded matrices that arrive through the reference-compatible constructors (CSC lower triangle, general CSR / CSC)
    n = 3001
    rng = np.random.default_rng(5)
    diags = [rng.uniform(-1, 1, n - abs(k)) for k in (0, 1, 4, 77)]
    L = sp.diags(diags, [0, -1, -4, -77], format="csc")
    S = (L + sp.tril(L, -1).T).tocsr()
    S.sort_indices()
    x = rand_x(n, 4)
    ref = O.Op.csr(n, n, S.indptr, S.indices, S.data).perform_op(x)
    op = sa.SparseSymMatProd(L, ctx=ctx)
    assert op.offset_codes() == 7
    y_codes, y_plain = _both_formats(op, x)
    assert np.array_equal(y_codes, ref) and np.array_equal(y_plain, ref)
    # rows with holes (entries removed at random keep the diagonal dictionary but make the rows ragged)
    G = sp.diags([rng.uniform(-1, 1, n - abs(k)) for k in (-300, -2, 0, 1, 9)], [-300, -2, 0, 1, 9], format="coo")
    keep = rng.uniform(size=G.nnz) < 0.6
    G = sp.coo_matrix((G.data[keep], (G.row[keep], G.col[keep])), shape=(n, n))
    for fmt in ("csr", "csc"):
        M = G.asformat(fmt)
        M.sort_indices()
        gop = sa.SparseGenMatProd(M, ctx=ctx)
        assert 0 < gop.offset_codes() <= 5 and gop.spmv_format() == 1  # 60 % filled diagonals: too sparse for diagonal storage
        Mr = M.tocsr()
        Mr.sort_indices()
        yc, yp = _both_formats(gop, x)
        assert np.array_equal(yc, yp)
        assert np.array_equal(yc, O.Op.csr(n, n, Mr.indptr, Mr.indices, Mr.data).perform_op(x))


def test_offset_codes_with_rows_longer_than_the_lds_chunk_and_rectangular_shapes(ctx):
    # dense 120 x 120 stored as sparse: 239 diagonals, 14400 entries in one row block -> several LDS chunks
    rng = np.random.default_rng(8)
    D = sp.csr_matrix(rng.uniform(-1, 1, (120, 120)))
    op = sa.SparseGenMatProd(D, ctx=ctx)
    assert op.offset_codes() == 239 and op.spmv_format() == 1  # more than 32 diagonals
    x = rand_x(120, 6)
    yc, yp = _both_formats(op, x)
    assert np.array_equal(yc, yp) and np.array_equal(yc, O.Op.csr(120, 120, D.indptr, D.indices, D.data).perform_op(x))
    # 129 x 129 dense has 257 diagonals: one more than the dictionary holds
    assert sa.SparseGenMatProd(sp.csr_matrix(rng.uniform(-1, 1, (129, 129))), ctx=ctx).offset_codes() == 0
    # rectangular band (the SVD operators): 700 x 2000 and its transpose
    R = sp.diags([rng.uniform(-1, 1, 700)] * 4, [0, 3, 650, 1299], shape=(700, 2000), format="csr")
    R.sort_indices()
    rop = sa.SparseGenMatProd(R, ctx=ctx)
    assert rop.offset_codes() == 4 and rop.spmv_format() == 2
    xr = rand_x(2000, 7)
    yc, yp = _both_formats(rop, xr)
    assert np.array_equal(yc, yp) and np.array_equal(yc, O.Op.csr(700, 2000, R.indptr, R.indices, R.data).perform_op(xr))
    Rt = R.T.tocsr()
    Rt.sort_indices()
    top = sa.SparseGenMatProd(Rt, ctx=ctx)
    assert top.offset_codes() == 4 and top.spmv_format() == 1  # 2000 rows, 700 entries per diagonal: under 3/4 full
    xt = rand_x(700, 8)
    yc, yp = _both_formats(top, xt)
    assert np.array_equal(yc, yp) and np.array_equal(yc, O.Op.csr(2000, 700, Rt.indptr, Rt.indices, Rt.data).perform_op(xt))


# ---- ragged rows, empty row-blocks -------------------------------------------------------------------------------
def test_ragged_rows_and_empty_blocks_in_both_index_formats(ctx):
    # random row lengths 0..14, a run of 600 empty rows (two empty 256-row blocks), empty first and last rows;
    # general (non-symmetric) matrix with too many diagonals for offset codes -> int32 indices
    n = 5000
    rng = np.random.default_rng(12)
    lens = rng.integers(0, 15, n)
    lens[1500:2100] = 0
    lens[0] = lens[-1] = 0
    rows = np.repeat(np.arange(n), lens)
    cols = rng.integers(0, n, rows.size)
    A = sp.coo_matrix((rng.uniform(-1, 1, rows.size), (rows, cols)), shape=(n, n)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    op = sa.SparseGenMatProd(A, ctx=ctx)
    assert op.offset_codes() == 0
    x = rand_x(n, 13)
    assert np.array_equal(op.perform_op(x), O.Op.csr(n, n, A.indptr, A.indices, A.data).perform_op(x))
    # the same on few diagonals -> offset codes, again with empty blocks
    B = sp.diags([rng.uniform(-1, 1, n - abs(k)) for k in (-40, -1, 0, 2, 300)], [-40, -1, 0, 2, 300], format="lil")
    B[1500:2100, :] = 0
    B[0, :] = 0
    B = B.tocsr()
    B.eliminate_zeros()
    B.sort_indices()
    bop = sa.SparseGenMatProd(B, ctx=ctx)
    assert 0 < bop.offset_codes() <= 5
    yc, yp = _both_formats(bop, x)
    assert np.array_equal(yc, yp)
    assert np.array_equal(yc, O.Op.csr(n, n, B.indptr, B.indices, B.data).perform_op(x))


def test_fused_epilogue_is_identical_in_every_storage_format():
    # the Lanczos epilogue (w -= beta v_prev, alpha partials) rides on the SpMV: a full solve must not depend on the format
    n = 200_000
    res = []
    for fmt in (0, 1, 2):
        op = sa.SparseSymMatProd.synth_band(n)
        op.set_spmv_format(fmt)
        assert op.spmv_format() == fmt
        eigs = sa.SymEigsSolver(op, 6, 20)
        eigs.set_orth_mode("reference")  # the one-sweep steps on diagonal storage divide the row sums instead of x (csr.hpp
        eigs.init()                      # post_scale_state): the same numbers to rounding, not to the bit — checked below
        eigs.compute(sa.SortRule.LargestMagn, tol=1e-11)
        res.append((eigs.eigenvalues(), eigs.num_operations()))
    for ev, nops in res[1:]:
        assert np.array_equal(ev, res[0][0]) and nops == res[0][1]
    one = []
    for fmt in (0, 1, 2):
        op = sa.SparseSymMatProd.synth_band(n)
        op.set_spmv_format(fmt)
        eigs = sa.SymEigsSolver(op, 6, 20)  # the default: one-sweep steps
        eigs.init()
        eigs.compute(sa.SortRule.LargestMagn, tol=1e-11)
        assert eigs.orth_info()["mode"] == "onesweep"
        one.append((eigs.eigenvalues(), eigs.num_operations()))
    assert np.array_equal(one[0][0], one[1][0]) and one[0][1] == one[1][1]  # the two CSR formats: the same kernels around them
    for ev, nops in one:
        assert np.abs(ev - res[0][0]).max() < 1e-10 and abs(nops - res[0][1]) <= 20


def test_diagonal_storage_needs_sorted_rows_without_duplicates(ctx):
    # through the raw C ABI (the Python classes sort and merge): a row with a duplicated entry, or with its columns out
    # of order, must keep the CSR kernels, whose storage-order sum is what the reference's row-major product does
    import ctypes as C

    n = 600
    T = sp.diags([np.ones(n - 1), np.ones(n), np.ones(n - 1)], [-1, 0, 1], format="csr")
    T.sort_indices()
    rp = T.indptr.astype(np.int32)
    vals = np.random.default_rng(1).uniform(-1, 1, T.nnz)
    x = rand_x(n, 2)
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    for kind, want in (("sorted", 2), ("swapped", 1), ("duplicate", 1)):
        ci = T.indices.astype(np.int32).copy()
        p = rp[10]
        if kind == "swapped":
            ci[p], ci[p + 1] = ci[p + 1], ci[p]
        if kind == "duplicate":
            ci[p + 1] = ci[p]
        h = C.c_void_p()
        sa.check(sa.lib().mispec_csr_upload(ctx.h, n, n, rp.ctypes.data_as(ip), ci.ctypes.data_as(ip), vals.ctypes.data_as(dp), C.byref(h)))
        assert sa.lib().mispec_csr_spmv_format(h) == want, kind
        y = np.empty(n)
        sa.check(sa.lib().mispec_spmv_host(h, x.ctypes.data_as(dp), y.ctypes.data_as(dp)))
        assert np.array_equal(y, O.Op.csr(n, n, rp, ci, vals).perform_op(x)), kind
        sa.check(sa.lib().mispec_csr_destroy(h))


@pytest.mark.parametrize("offsets,fmt", [
    (tuple(range(1, 4)) + tuple(range(500, 504)) + tuple(range(9000, 9003)), 2),   # 21 diagonals, 5 clusters: windows, 3 groups
    (tuple(1000 * k for k in range(1, 16)), 2),                                    # 31 diagonals, 31 clusters: direct x loads, 4 groups
    (tuple(range(1, 16)), 2),                                                      # 31 diagonals in one cluster: windows, 4 groups
    (tuple(7 * k for k in range(1, 17)), 1),                                       # 33 diagonals: offset codes only
])
def test_diagonal_storage_with_many_diagonals(ctx, offsets, fmt):
    n = 40_000
    op = sa.SparseSymMatProd.synth_band(n, offsets=offsets, ctx=ctx)
    assert op.offset_codes() == 2 * len(offsets) + 1 and op.spmv_format() == fmt
    x = rand_x(n, 17)
    y_codes, y_plain = _both_formats(op, x)
    assert np.array_equal(y_codes, y_plain)
    rp, ci, v = O.synth_band_csr(n, offsets=offsets)
    assert np.array_equal(y_codes, O.Op.csr(n, n, rp, ci, v).perform_op(x))
