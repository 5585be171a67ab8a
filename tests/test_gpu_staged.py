"""The staged SpMV (spectra_amd/csrc/staged.hip, storage format 4: products column block by column block with x in LDS, row sums
bin by bin with y in LDS) on the GPU: BIT-IDENTICAL to the CSR kernel, to the tiles and to the oracle's row-dot, fused Lanczos
epilogue included (whole solves equal to the last bit)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import oracle as O
import spectra_amd as sa
from test_gpu_tiles import device_spmv, m_rand

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def every_format(monkeypatch):
    """Staged format forced for every matrix of this module (the automatic choice takes it for scattered patterns only), tiles
    built next to it so that the formats can be compared on one operator."""
    monkeypatch.setenv("MISPEC_SPMV_STAGED", "1")
    monkeypatch.setenv("MISPEC_SPMV_TILES", "1")


def test_scattered_patterns_get_the_staged_format_by_default(ctx, monkeypatch):
    monkeypatch.delenv("MISPEC_SPMV_STAGED")
    monkeypatch.delenv("MISPEC_SPMV_TILES")
    A = m_rand(300_001)
    op = sa.SparseSymMatProd(sp.tril(A).tocsc(), ctx=ctx)
    assert op.reordering() == "none" and op.spmv_format() == 4 and op.staged_info()["bins"] > 0 and op.tiles_info()["segments"] == 0
    x = np.random.default_rng(2).standard_normal(300_001)
    assert np.array_equal(device_spmv(op, x), O.Op.csr(300_001, 300_001, A.indptr, A.indices, A.data).perform_op(x))


def test_heavy_rows_make_the_automatic_choice_decline_the_staged_format(ctx, monkeypatch):
    # three dense rows among scattered ones would give one nearly empty batch per 8 of their entries (serial barrier
    # rounds of one workgroup).  The automatic choice keeps another kernel; the product stays the CSR row-dot bit for bit.
    monkeypatch.delenv("MISPEC_SPMV_STAGED")
    monkeypatch.delenv("MISPEC_SPMV_TILES")
    n = 300_001
    A = m_rand(n).tolil()
    rng = np.random.default_rng(4)
    for r in (11, 150_000, n - 2):
        A[r, :] = rng.uniform(-1, 1, n)
    A = A.tocsr()
    A.sort_indices()
    op = sa.SparseGenMatProd(A, ctx=ctx)
    assert op.spmv_format() != 4 and op.staged_info()["bins"] == 0
    x = np.random.default_rng(2).standard_normal(n)
    assert np.array_equal(device_spmv(op, x), O.Op.csr(n, n, A.indptr, A.indices, A.data).perform_op(x))


@pytest.mark.parametrize("n", [300_001, 1_000_000])
def test_staged_product_is_bit_exact(ctx, n):
    A = m_rand(n)
    op = sa.SparseSymMatProd(sp.tril(A).tocsc(), ctx=ctx)
    info = op.staged_info()
    assert 0 < info["bins"] <= max(512, (n + 255) // 256) and op.spmv_format() == 4
    ref_op = O.Op.csr(n, n, A.indptr, A.indices, A.data)
    for seed in (0, 1):
        x = np.random.default_rng(seed).standard_normal(n) * (1.0 if seed == 0 else np.exp(np.random.default_rng(9).uniform(-15, 15, n)))
        y_ref = ref_op.perform_op(x)
        assert np.array_equal(device_spmv(op, x), y_ref)
        for fmt in (3, 0):
            op.set_spmv_format(fmt)
            assert op.spmv_format() == fmt and np.array_equal(device_spmv(op, x), y_ref)
        op.set_spmv_format(-1)
    assert np.array_equal(op.perform_op(np.ones(n)), ref_op.perform_op(np.ones(n)))  # host-pointer path


def test_solves_with_the_staged_format_and_with_csr_are_identical(ctx):
    n = 400_000
    A = m_rand(n, seed=4)
    op = sa.SparseSymMatProd(sp.tril(A).tocsc(), ctx=ctx)
    assert op.spmv_format() == 4
    out = []
    for fmt in (4, 3, 0):
        op.set_spmv_format(fmt)
        for mode in ("reference", "onesweep"):
            e = sa.SymEigsSolver(op, 6, 24)
            e.set_orth_mode(mode)
            e.init()
            nconv = e.compute(sa.SortRule.LargestAlge, 40, 1e-9)
            out.append((fmt, mode, nconv, e.num_operations(), e.num_iterations(), e.eigenvalues().tobytes()))
    op.set_spmv_format(-1)
    for mode in ("reference", "onesweep"):
        runs = [o[2:] for o in out if o[1] == mode]
        assert runs[0] == runs[1] == runs[2]  # the fused epilogue writes the same alpha records: bit-identical solves


def test_staged_on_clustered_dense_row_and_rectangular_matrices(ctx):
    rng = np.random.default_rng(3)
    for nr, nc in ((5000, 5000), (9001, 300_000), (8192, 140_000), (20_000, 9000)):
        rows, cols = [], []
        for r in range(0, nr, 2):
            k = int(rng.integers(1, 30))
            cs = np.sort(rng.choice(min(nc, 3000), k, replace=False)) + (nc - min(nc, 3000)) * int(rng.integers(0, 2))
            rows += [r] * k
            cols += cs.tolist()
        rows += [1] * nc  # one completely dense row: batches of exactly 1024 entries, every rank of the field
        cols += list(range(nc))
        A = sp.coo_matrix((rng.uniform(-1, 1, len(rows)), (rows, cols)), shape=(nr, nc)).tocsr()
        A.sort_indices()
        op = sa.SparseGenMatProd(A, ctx=ctx, reorder="none")
        assert op.spmv_format() == 4
        x = rng.standard_normal(nc)
        assert np.array_equal(device_spmv(op, x), O.Op.csr(nr, nc, A.indptr, A.indices, A.data).perform_op(x))
