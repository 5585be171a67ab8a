"""The column-blocked tile SpMV (spectra_amd/csrc/tiles.hip, storage format 3) on the GPU: built at ingest for scattered
patterns (SURVEY.md 8d "M-rand"), BIT-IDENTICAL to the CSR kernel and to the oracle's row-dot, fused Lanczos epilogue
included (whole solves equal to the last bit)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import oracle as O
import spectra_amd as sa

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def tiles_are_the_scattered_format(monkeypatch):
    """Since round 4 scattered patterns get the staged format (tests/test_gpu_staged.py); this module tests the tiles."""
    monkeypatch.setenv("MISPEC_SPMV_STAGED", "0")


def m_rand(n, seed=1):
    rng = np.random.default_rng(seed)
    r = np.repeat(np.arange(n), 7)
    U = sp.coo_matrix((rng.uniform(-0.5, 0.5, r.size), (r, rng.integers(0, n, r.size))), shape=(n, n)).tocsr()
    U.sum_duplicates()
    A = (U + U.T + sp.diags(rng.uniform(-0.5, 0.5, n))).tocsr()
    A.sort_indices()
    return A


def device_spmv(op, x):
    import torch

    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    yd = torch.empty(op.local_rows() + 2, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    op.spmv_device(xd.data_ptr(), yd.data_ptr())
    op.ctx.sync()
    return yd[: op.local_rows()].cpu().numpy()


@pytest.mark.parametrize("n", [300_001, 1_000_000])
def test_scattered_matrix_gets_tiles_and_the_product_is_bit_exact(ctx, n):
    A = m_rand(n)
    op = sa.SparseSymMatProd(sp.tril(A).tocsc(), ctx=ctx)
    assert op.reordering() == "none" and op.spmv_format() == 3          # the expander is not reordered; tiles are built
    ref_op = O.Op.csr(n, n, A.indptr, A.indices, A.data)
    for seed in (0, 1):
        x = np.random.default_rng(seed).standard_normal(n) * (1.0 if seed == 0 else np.exp(np.random.default_rng(9).uniform(-15, 15, n)))
        y_ref = ref_op.perform_op(x)
        y3 = device_spmv(op, x)
        assert np.array_equal(y3, y_ref)
        op.set_spmv_format(0)
        assert op.spmv_format() == 0 and np.array_equal(device_spmv(op, x), y_ref)
        op.set_spmv_format(-1)
    assert np.array_equal(op.perform_op(np.ones(n)), ref_op.perform_op(np.ones(n)))  # host-pointer path


def test_solves_with_tiles_and_with_csr_are_identical(ctx):
    n = 400_000
    A = m_rand(n, seed=4)
    op = sa.SparseSymMatProd(sp.tril(A).tocsc(), ctx=ctx)
    assert op.spmv_format() == 3
    out = []
    for fmt in (3, 0):
        op.set_spmv_format(fmt)
        e = sa.SymEigsSolver(op, 6, 24)
        e.init()
        nconv = e.compute(sa.SortRule.LargestAlge, 40, 1e-9)
        out.append((nconv, e.num_operations(), e.num_iterations(), e.eigenvalues().tobytes()))
    op.set_spmv_format(-1)
    assert out[0] == out[1]      # the fused epilogue writes the same alpha records: bit-identical solves
    fac = sa.Factorization(op, 10, True)
    fac.init_random(0)
    fac.factorize_from(1, 10)
    ofac = O.Factorization(O.Op.csr(n, n, A.indptr, A.indices, A.data), 10, True)
    ofac.init(O.simple_random(n, 0))
    ofac.factorize_from(1, 10)
    assert np.abs(fac.matrix_H() - ofac.matrices()[1]).max() < 1e-10


def test_forced_tiles_on_small_clustered_and_rectangular_matrices(ctx):
    # MISPEC_SPMV_TILES=1 builds the format for any matrix: long runs (several passes), empty rows, ragged last segment,
    # rectangular shape
    rng = np.random.default_rng(3)
    old = os.environ.get("MISPEC_SPMV_TILES")
    os.environ["MISPEC_SPMV_TILES"] = "1"
    try:
        for nr, nc in ((5000, 5000), (9001, 300_000), (4096, 140_000)):
            rows, cols = [], []
            for r in range(0, nr, 2):
                k = int(rng.integers(1, 30))
                cs = np.sort(rng.choice(min(nc, 3000), k, replace=False)) + (nc - min(nc, 3000)) * int(rng.integers(0, 2))
                rows += [r] * k
                cols += cs.tolist()
            A = sp.coo_matrix((rng.uniform(-1, 1, len(rows)), (rows, cols)), shape=(nr, nc)).tocsr()
            A.sort_indices()
            op = sa.SparseGenMatProd(A, ctx=ctx, reorder="none")
            assert op.spmv_format() == 3
            x = rng.standard_normal(nc)
            assert np.array_equal(device_spmv(op, x), O.Op.csr(nr, nc, A.indptr, A.indices, A.data).perform_op(x))
    finally:
        if old is None:
            del os.environ["MISPEC_SPMV_TILES"]
        else:
            os.environ["MISPEC_SPMV_TILES"] = old
