"""Dense operators (MatOp/DenseSymMatProd.h, MatOp/DenseGenMatProd.h) and user operators on device pointers, through
the C ABI, against the CPU oracle.

The reference's own solver tests run mostly on dense matrices (test/SymEigs.cpp:100-131, test/GenEigs.cpp:110-146);
their `Matrix::Random` fixtures are Eigen-internal and cannot be regenerated, so the same shapes are used with seeded
numpy matrices and the bar is the reference's: ||AU - UD||_inf <= 1e-9, eigenvalues equal to the oracle's / LAPACK's.
"""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

import oracle as O
import spectra_amd as sa
from helpers import RULES_SYM, sparse_fixture, wanted_by_rule

pytestmark = pytest.mark.gpu

DENSE_CASES = [(10, 3, 6), (100, 10, 20), (100, 10, 30), (1000, 20, 50)]  # test/SymEigs.cpp:100-131


def sym_dense(n, seed):
    """A non-symmetric input whose LOWER triangle defines the operator (test/SymEigs.cpp:19-23 builds M + M')."""
    rng = np.random.default_rng(seed)
    M = rng.uniform(-1, 1, (n, n))
    S = np.tril(M) + np.tril(M, -1).T
    return M, S


@pytest.mark.parametrize("n", [1, 2, 7, 64, 129, 500])
@pytest.mark.parametrize("order", ["F", "C"])
def test_dense_sym_operator_members(ctx, n, order):
    # test/DenseSymMatProd.cpp:37-58: op * M == mat.selfadjointView * M, op(i, j) == mat(i, j); only `uplo` is read
    M, S = sym_dense(n, n)
    Min = np.array(M, order=order)
    op = sa.DenseSymMatProd(Min, ctx=ctx)
    assert (op.rows(), op.cols()) == (n, n)
    x = np.random.default_rng(1).uniform(-1, 1, n)
    assert np.abs(op.perform_op(x) - S @ x).max() <= 1e-13 * max(1.0, n)
    X = np.random.default_rng(2).uniform(-1, 1, (n, 3))
    assert np.allclose(op @ X, S @ X, rtol=0, atol=1e-12)
    i, j = min(5, n - 1), min(2, n - 1)
    assert op(i, j) == S[i, j] and op(j, i) == S[j, i]
    up = sa.DenseSymMatProd(Min, uplo="U", ctx=ctx)
    Su = np.triu(M) + np.triu(M, 1).T
    assert np.abs(up.perform_op(x) - Su @ x).max() <= 1e-13 * max(1.0, n)
    # deterministic: the same launch twice gives the same bits
    assert np.array_equal(op.perform_op(x), op.perform_op(x))


@pytest.mark.parametrize("shape", [(1, 1), (5, 9), (130, 77), (300, 300)])
def test_dense_gen_operator_members(ctx, shape):
    # test/DenseGenMatProd.cpp:37-57
    rng = np.random.default_rng(shape[0])
    M = rng.uniform(-1, 1, shape)
    for Min in (np.asfortranarray(M), np.ascontiguousarray(M)):
        op = sa.DenseGenMatProd(Min, ctx=ctx)
        assert (op.rows(), op.cols()) == shape
        x = rng.uniform(-1, 1, shape[1])
        assert np.abs(op.perform_op(x) - M @ x).max() <= 1e-13 * shape[1]
        X = rng.uniform(-1, 1, (shape[1], 4))
        assert np.allclose(op @ X, M @ X, rtol=0, atol=1e-12)
        assert op(shape[0] - 1, shape[1] - 1) == M[-1, -1] and op(0, shape[1] - 1) == M[0, -1]


def test_dense_error_behaviour(ctx):
    with pytest.raises(ValueError):
        sa.DenseSymMatProd(np.zeros((3, 4)), ctx=ctx)  # must be square
    op = sa.DenseGenMatProd(np.eye(5, 7), ctx=ctx)
    with pytest.raises(ValueError):
        sa.GenEigsSolver(op, 2, 4).init()  # the eigen solvers need a square operator
    with pytest.raises(ValueError):
        op.perform_op(np.zeros(5))


@pytest.mark.parametrize("n,k,m", DENSE_CASES)
@pytest.mark.parametrize("rule", RULES_SYM)
def test_sym_eigs_on_dense_matrices_all_rules(ctx, n, k, m, rule):
    M, S = sym_dense(n, 100 + n)
    eigs = sa.SymEigsSolver(sa.DenseSymMatProd(M, ctx=ctx), k, m)
    eigs.init()
    nconv = eigs.compute(sa.SortRule[rule])
    assert eigs.info() == sa.CompInfo.Successful and nconv == k
    evals, evecs = eigs.eigenvalues(), eigs.eigenvectors()
    assert np.abs(S @ evecs - evecs * evals).max() < 1e-9  # test/SymEigs.cpp:64
    assert np.abs(np.sort(evals) - wanted_by_rule(np.linalg.eigvalsh(S), rule, k)).max() < 1e-9
    ref = O.SymEigsSolver(O.Op.dense_sym(S), k, m)
    ref.init()
    ref.compute(getattr(O, rule))
    assert np.abs(ref.eigenvalues() - evals).max() < 1e-9
    assert abs(eigs.num_operations() - ref.num_operations()) <= max(3 * m, 0.15 * ref.num_operations())


@pytest.mark.parametrize("n,k,m", [(10, 3, 6), (100, 10, 30), (500, 12, 40)])
def test_gen_eigs_on_dense_matrices(ctx, n, k, m):
    # test/GenEigs.cpp:110-146 shapes; residual ||AU - UD||_inf <= 1e-9 on the complex pairs
    A = np.random.default_rng(7 + n).uniform(-1, 1, (n, n))
    eigs = sa.GenEigsSolver(sa.DenseGenMatProd(A, ctx=ctx), k, m)
    eigs.init()
    nconv = eigs.compute(sa.SortRule.LargestMagn)
    assert eigs.info() == sa.CompInfo.Successful and nconv >= k - 1
    evals, evecs = eigs.eigenvalues(), eigs.eigenvectors()
    assert np.abs(A @ evecs - evecs * evals).max() < 1e-9
    full = np.linalg.eigvals(A)
    full = full[np.argsort(-np.abs(full))]
    assert np.abs(np.sort(np.abs(evals))[::-1] - np.abs(full[:len(evals)])).max() < 1e-9
    ref = O.GenEigsSolver(O.Op.dense_gen(A), k, m)
    ref.init()
    ref.compute(O.LargestMagn)
    assert np.abs(np.sort_complex(ref.eigenvalues()) - np.sort_complex(evals)).max() < 1e-9


def test_factorisation_on_a_dense_operator_equals_the_sparse_one(ctx):
    # the same matrix as a dense and as a CSR operator: identical Lanczos identities, H equal to rounding
    A, S = sparse_fixture(100, 0.1)
    fd = sa.Factorization(sa.DenseSymMatProd(S.toarray(), ctx=ctx), 20)
    fs = sa.Factorization(sa.SparseSymMatProd(A, ctx=ctx), 20)
    v0 = np.random.default_rng(3).uniform(-0.5, 0.5, 100)
    for f in (fd, fs):
        f.init(v0)
        f.factorize_from(1, 20)
    V, H, fr = fd.matrix_V(), fd.matrix_H(), fd.vector_f()
    E = np.zeros((100, 20))
    E[:, -1] = fr
    assert np.abs(S @ V - V @ H - E).max() < 1e-12       # test/Arnoldi.cpp:61-63
    assert np.abs(V.T @ V - np.eye(20)).max() < 1e-12
    assert np.abs(H - fs.matrix_H()).max() < 1e-10
    assert fd.num_operations() == fs.num_operations() == 21  # init: 2 (v = A v0, w = A v), then 19 steps


@pytest.mark.parametrize("orth", ["onesweep", "reference"])
def test_user_operator_on_device_pointers(ctx, orth, monkeypatch):
    monkeypatch.setenv("MISPEC_ORTH", orth)
    # y = A x through the library's own SpMV called from the callback with DEVICE pointers: the factorisation never
    # leaves HBM, and the solve must agree with the bound device matrix (same kernels apart from the fused epilogue)
    n, k, m = 1000, 20, 50
    A, S = sparse_fixture(n, 0.01)
    mat = sa.SparseSymMatProd(A, ctx=ctx)
    calls = []

    def apply(x_ptr, y_ptr, stream):
        calls.append(stream)
        mat.spmv_device(x_ptr, y_ptr)

    eigs = sa.SymEigsSolver(sa.DeviceOp(n, apply, ctx=ctx), k, m)
    eigs.init()
    nconv = eigs.compute(sa.SortRule.LargestAlge)
    assert eigs.info() == sa.CompInfo.Successful and nconv == k
    evals, evecs = eigs.eigenvalues(), eigs.eigenvectors()
    assert np.abs(S @ evecs - evecs * evals).max() < 1e-9
    assert len(calls) == eigs.num_operations()
    # device-pointer operators take the device-driven steps, in the default mode the one-sweep ones
    info = eigs.orth_info()
    assert info["mode"] == orth and (info["lagged_steps"] > 0) == (orth == "onesweep")
    ref = sa.SymEigsSolver(mat, k, m)
    ref.init()
    ref.compute(sa.SortRule.LargestAlge)
    assert np.abs(ref.eigenvalues() - evals).max() < 1e-10

    # a failing callback surfaces as an error, not as a wrong answer
    def broken(x_ptr, y_ptr, stream):
        raise RuntimeError("boom")

    bad = sa.SymEigsSolver(sa.DeviceOp(n, broken, ctx=ctx), k, m)
    with pytest.raises(RuntimeError):
        bad.init()

    # general solver on a device operator
    G = sp.random(300, 300, density=0.05, random_state=5, format="csr")
    G.data[:] = np.random.default_rng(5).uniform(-1, 1, G.nnz)
    G.sort_indices()
    gm = sa.SparseGenMatProd(G, ctx=ctx)
    ge = sa.GenEigsSolver(sa.DeviceOp(300, lambda x, y, s: gm.spmv_device(x, y), ctx=ctx), 6, 30)
    ge.init()
    assert ge.compute(sa.SortRule.LargestMagn) >= 5
    ev, U = ge.eigenvalues(), ge.eigenvectors()
    assert np.abs(G @ U - U * ev).max() < 1e-9


def test_dense_gemv_rate_at_scale(ctx):
    # size-independent property at a size the oracle cannot handle in seconds (n = 16384: 2.1 GB of matrix): linearity
    import torch

    n = 16384
    g = torch.Generator(device="cpu").manual_seed(0)
    M = (torch.rand((n, n), dtype=torch.float64, generator=g) - 0.5).numpy()
    op = sa.DenseGenMatProd(M, ctx=ctx)
    x = torch.rand(n, dtype=torch.float64, device="cuda") - 0.5
    y = torch.rand(n, dtype=torch.float64, device="cuda") - 0.5
    z = 0.5 * x - 2.0 * y
    Ax, Ay, Az = (torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(3))
    torch.cuda.synchronize()
    op.gemv_device(x.data_ptr(), Ax.data_ptr())
    op.gemv_device(y.data_ptr(), Ay.data_ptr())
    op.gemv_device(z.data_ptr(), Az.data_ptr())
    ctx.sync()
    assert (Az - (0.5 * Ax - 2.0 * Ay)).abs().max().item() < 1e-10
    rows = [0, 1, n // 2, n - 1]
    xh = x.cpu().numpy()
    assert np.abs(Ax.cpu().numpy()[rows] - M[rows] @ xh).max() < 1e-10
    ms = op.gemv_time(x.data_ptr(), Ax.data_ptr(), 20)
    gbps = op.algorithmic_bytes() / (ms * 1e-3) / 1e9
    print(f"dense GEMV n={n}: {ms:.3f} ms, {gbps:.0f} GB/s")
    assert gbps > 1500  # HBM-bound kernel; anything far below means the row streaming broke


def test_dense_shift_solve_and_cholesky_operators(ctx):
    # MatOp/DenseSymShiftSolve.h, DenseGenRealShiftSolve.h, DenseCholesky.h: dense inputs through the device
    # factorisations; shapes of test/SymEigsShift.cpp:112-158, test/GenEigsRealShift.cpp:110-146, test/SymGEigsCholesky.cpp
    n, k, m = 100, 10, 30
    M, S = sym_dense(n, 77)
    sigma = 0.5
    op = sa.DenseSymShiftSolve(M, ctx=ctx)
    op.set_shift(sigma)
    x = np.random.default_rng(3).uniform(-1, 1, n)
    y = op.perform_op(x)
    assert np.abs((S - sigma * np.eye(n)) @ y - x).max() < 1e-10
    eigs = sa.SymEigsShiftSolver(sa.DenseSymShiftSolve(M, ctx=ctx), k, m, sigma)
    eigs.init()
    assert eigs.compute(sa.SortRule.LargestMagn) == k
    evals, evecs = eigs.eigenvalues(), eigs.eigenvectors()
    assert np.abs(S @ evecs - evecs * evals).max() < 1e-9
    full = np.linalg.eigvalsh(S)
    near = full[np.argsort(np.abs(full - sigma))[:k]]
    assert np.abs(np.sort(evals) - np.sort(near)).max() < 1e-9
    # general real shift
    G = np.random.default_rng(5).uniform(-1, 1, (n, n))
    ge = sa.GenEigsRealShiftSolver(sa.DenseGenRealShiftSolve(G, ctx=ctx), k, m, 1.0)
    ge.init()
    assert ge.compute(sa.SortRule.LargestMagn) >= k - 1
    ev, U = ge.eigenvalues(), ge.eigenvectors()
    assert np.abs(G @ U - U * ev).max() < 1e-8
    # Cholesky mode with a sparse A and a dense B
    A, Sa = sparse_fixture(n, 0.1)
    Bd = (Sa.T @ Sa).toarray() + 0.1 * np.eye(n)
    B = sa.DenseCholesky(Bd, ctx=ctx)
    assert B.info() == sa.CompInfo.Successful
    g = sa.SymGEigsSolver(sa.SparseSymMatProd(A, ctx=ctx), B, 10, 20, mode="Cholesky")
    g.init()
    assert g.compute(sa.SortRule.LargestAlge) == 10
    lam, X = g.eigenvalues(), g.eigenvectors()
    assert np.abs(Sa @ X - Bd @ X * lam).max() < 1e-9
    with pytest.raises(ValueError):
        sa.DenseCholesky(np.zeros((3, 4)), ctx=ctx)


@pytest.mark.parametrize("n", [61, 127])
def test_odd_sized_dense_factor_paths(ctx, n):
    # odd n: the explicit inverses have an odd row stride, which takes the GEMV kernel's scalar (non-16-byte) path
    M, S = sym_dense(n, 5)
    x = np.random.default_rng(6).uniform(-1, 1, n)
    op = sa.DenseSymShiftSolve(M, ctx=ctx)
    op.set_shift(0.25)
    assert np.abs((S - 0.25 * np.eye(n)) @ op.perform_op(x) - x).max() < 1e-10
    G = np.random.default_rng(7).uniform(-1, 1, (n, n))
    g = sa.DenseGenComplexShiftSolve(G, ctx=ctx)
    g.set_shift(0.1, 0.4)
    assert np.abs(g.perform_op(x) - np.linalg.solve(G - (0.1 + 0.4j) * np.eye(n), x).real).max() < 1e-10
    B = S @ S.T + np.eye(n)
    chol = sa.DenseCholesky(B, ctx=ctx)
    L = np.linalg.cholesky(B)
    assert np.abs(chol.lower_triangular_solve(x) - np.linalg.solve(L, x)).max() < 1e-10
    assert np.abs(chol.upper_triangular_solve(x) - np.linalg.solve(L.T, x)).max() < 1e-10
