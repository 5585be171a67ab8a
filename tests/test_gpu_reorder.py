"""Reordered matrices (P A P' by reverse Cuthill-McKee at ingest, spectra_amd/csrc/reorder.hip) on the GPU.

* the SpMV on the permuted CSR is BIT-EXACT against the oracle's row-dot on the same permuted CSR;
* everything handed out keeps the caller's index order (products, coefficients, downloads, eigenvectors);
* a solve on the reordered matrix gives the eigenvalues of the unreordered solve (<= 1e-12) and eigenvectors in the
  caller's order; GenEigsSolver too."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle as O
import spectra_amd as sa
from test_host_reorder import bandwidth, stencil7

pytestmark = pytest.mark.gpu


def shuffled_stencil(m, seed=0):
    A = stencil7(m)
    n = A.shape[0]
    rng = np.random.default_rng(seed)
    A = A.tocsr()
    A.data[:] = rng.uniform(-0.5, 0.5, A.nnz)
    A = sp.tril(A) + sp.tril(A, -1).T
    p = rng.permutation(n)
    B = A.tocsr()[p][:, p].tocsr()
    B.sort_indices()
    return B


def test_reordered_spmv_is_bit_exact_on_the_permuted_csr_and_keeps_the_callers_order(ctx):
    B = shuffled_stencil(30)
    n = B.shape[0]
    op = sa.SparseSymMatProd(sp.tril(B).tocsc(), ctx=ctx, reorder="rcm")
    plain = sa.SparseSymMatProd(sp.tril(B).tocsc(), ctx=ctx, reorder="none")
    assert op.reordering() == "rcm" and plain.reordering() == "none"
    perm = op.permutation()
    assert sorted(perm.tolist()) == list(range(n))
    info = op.reordering_info()
    assert info["far_fraction_after"] <= info["far_fraction_before"]
    # the stored matrix is P B P' with sorted rows: the oracle's row-dot on exactly that CSR, on the permuted vector
    Bp = B[perm][:, perm].tocsr()
    Bp.sort_indices()
    assert bandwidth(Bp) < bandwidth(B) // 4
    x = np.random.default_rng(1).uniform(-1, 1, n)
    y = op.perform_op(x)                                     # caller's order in, caller's order out
    ref_perm = O.Op.csr(n, n, Bp.indptr, Bp.indices, Bp.data).perform_op(x[perm])
    assert np.array_equal(y[perm], ref_perm)                 # bit-exact on the permuted CSR
    y_plain = plain.perform_op(x)
    assert np.abs(y - y_plain).max() <= 1e-14 * max(1.0, np.abs(y_plain).max())  # another summation order only
    i, j = 17, int(B.indices[B.indptr[17]])
    assert op(i, j) == B[i, j] and op(j, i) == B[j, i]
    rp, ci, v = op.to_host_csr()
    assert np.array_equal(rp, B.indptr) and np.array_equal(ci, B.indices) and np.array_equal(v, B.data)
    X = np.random.default_rng(2).uniform(-1, 1, (n, 3))
    assert np.allclose(op @ X, B @ X, rtol=0, atol=1e-13)


@pytest.mark.parametrize("orth", ["reference", "onesweep"])
@pytest.mark.parametrize("rule", ["LargestMagn", "SmallestAlge", "BothEnds"])
def test_reordered_solve_equals_the_unreordered_one(ctx, rule, orth):
    # orth = "onesweep": the one-sweep steps and the fused restart on a factorisation that works in the stored (permuted) order
    B = shuffled_stencil(24, seed=3)
    n = B.shape[0]
    out = []
    for mode in ("none", "rcm"):
        op = sa.SparseSymMatProd(sp.tril(B).tocsc(), ctx=ctx, reorder=mode)
        eigs = sa.SymEigsSolver(op, 8, 24)
        eigs.set_orth_mode(orth)
        eigs.init()
        assert eigs.compute(sa.SortRule[rule], 1000, 1e-12) == 8
        ev, X = eigs.eigenvalues(), eigs.eigenvectors()
        assert np.abs(B @ X - X * ev).max() <= 1e-9             # eigenvectors are in the CALLER's order
        assert eigs.residuals().max() <= 1e-10
        assert eigs.eigenvectors(to_host=False) == 8
        if orth == "onesweep":
            info = eigs.orth_info()
            assert info["mode"] == "onesweep" and info["lagged_steps"] > 0 and info["fused_restarts"] > 0
        out.append((ev, X, eigs.num_operations()))
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-12
    assert np.abs(np.abs(np.sum(out[0][1] * out[1][1], axis=0)) - 1.0).max() <= 1e-8
    assert abs(out[0][2] - out[1][2]) <= 2 * 16                # the same algorithm up to rounding
    # and against the oracle on the caller's matrix
    ref = O.SymEigsSolver(O.Op.csr(n, n, B.indptr, B.indices, B.data), 8, 24)
    ref.init()
    assert ref.compute(getattr(O, rule), 1000, 1e-12) == 8
    assert np.abs(ref.eigenvalues() - out[1][0]).max() <= 1e-9


def test_user_start_vector_and_factorisation_outputs_keep_the_callers_order(ctx):
    B = shuffled_stencil(16, seed=5)
    n = B.shape[0]
    v0 = np.random.default_rng(7).uniform(-1, 1, n)
    res = []
    for mode in ("none", "rcm"):
        op = sa.SparseSymMatProd(sp.tril(B).tocsc(), ctx=ctx, reorder=mode)
        fac = sa.Factorization(op, 12, True)
        fac.init(v0)
        fac.factorize_from(1, 12)
        V, H, f = fac.matrix_V(), fac.matrix_H(), fac.vector_f()
        assert np.abs(V.T @ V - np.eye(12)).max() <= 1e-12
        assert np.abs(B @ V - V @ H - np.outer(f, np.eye(12)[11])).max() <= 1e-12   # A V = V H + f e' in the caller's order
        res.append((V, H))
    assert np.abs(res[0][1] - res[1][1]).max() <= 1e-11
    assert np.abs(np.abs(res[0][0][:, 0]) - np.abs(res[1][0][:, 0])).max() <= 1e-13


def test_reordered_general_matrix(ctx):
    # GenEigsSolver on a non-symmetric matrix in scattered order: pattern of A + A' drives the ordering
    m = 20
    A = stencil7(m).tocsr()
    rng = np.random.default_rng(11)
    A.data[:] = rng.uniform(-0.5, 0.5, A.nnz)
    n = A.shape[0]
    p = rng.permutation(n)
    B = A[p][:, p].tocsr()
    B.sort_indices()
    out = []
    for mode in ("none", "rcm"):
        op = sa.SparseGenMatProd(B, ctx=ctx, reorder=mode)
        assert op.reordering() == mode
        e = sa.GenEigsSolver(op, 6, 20)
        e.init()
        assert e.compute(sa.SortRule.LargestMagn, 1000, 1e-11) == 6
        ev, X = e.eigenvalues(), e.eigenvectors()
        assert (np.linalg.norm(B @ X - X * ev, axis=0) / np.linalg.norm(X, axis=0)).max() <= 1e-9
        assert e.residuals().max() <= 1e-9
        out.append(ev)
    for z in out[0]:
        assert np.abs(out[1] - z).min() <= 1e-10


def test_automatic_reordering_at_ingest(ctx):
    # 80^3 = 512000 rows: a shuffled stencil (half of its gathers further than the far window) is reordered without being
    # asked, a banded matrix and an expander are left alone
    B = shuffled_stencil(80, seed=2)
    op = sa.SparseSymMatProd(sp.tril(B).tocsc(), ctx=ctx)
    info = op.reordering_info()
    assert info["method"] == "rcm" and info["far_fraction_before"] > 0.25 and info["far_fraction_after"] == 0.0
    x = np.random.default_rng(0).uniform(-1, 1, B.shape[0])
    assert np.abs(op.perform_op(x) - B @ x).max() <= 1e-13
    band = sa.SparseSymMatProd(sp.tril(stencil7(80)).tocsc(), ctx=ctx)
    assert band.reordering() == "none"
    n = 300_000
    rng = np.random.default_rng(1)
    r = np.repeat(np.arange(n), 7)
    U = sp.coo_matrix((rng.uniform(-1, 1, r.size), (r, rng.integers(0, n, r.size))), shape=(n, n)).tocsr()
    S = (U + U.T).tocsr()
    rnd = sa.SparseSymMatProd(sp.tril(S).tocsc(), ctx=ctx)
    assert rnd.reordering() == "none" and rnd.reordering_info()["far_fraction_before"] > 0.25
    assert rnd.reorder("auto") is False


def test_davidson_and_regular_inverse_use_the_callers_diagonal_on_a_reordered_matrix(ctx, monkeypatch):
    # the stored matrix of a reordered operator is P A P', so its i-th diagonal entry is A(perm[i], perm[i]); the
    # Davidson preconditioner / unit start vectors and the Jacobi preconditioner of the CG solve work in the CALLER's order and
    # must not read that diagonal unpermuted.  A diagonal ramp makes the difference decisive: with the permuted diagonal
    # the preconditioner divides by the wrong entries and the iteration counts blow up.
    # 3-D stencil in its natural order (rows i, i + 1 are neighbours, so the solver's initial unit vectors — the rows with the
    # extreme diagonal entries — are coupled; on a matrix where they are not, the reference's DPR correction is 0 / 0 in its
    # first step and the solve ends with NumericalIssue, reordered or not), random symmetric couplings, a_ii = i + 1 as in the
    # reference's Davidson fixture (test/DavidsonSymEigs.cpp:46-67).  Forced RCM still renumbers it.
    S7 = stencil7(24).tocsr()
    S7.data[:] = np.random.default_rng(3).uniform(-0.5, 0.5, S7.nnz)
    B = (0.2 * (sp.tril(S7) + sp.tril(S7, -1).T)).tolil()
    n = B.shape[0]
    B.setdiag(np.arange(1.0, n + 1.0))
    B = B.tocsr()
    B.sort_indices()
    k = 5
    out = {}
    for mode in ("none", "rcm"):
        op = sa.SparseSymMatProd(sp.tril(B).tocsc(), ctx=ctx, reorder=mode)
        assert op.reordering() == mode
        eigs = sa.DavidsonSymEigsSolver(op, k)
        nconv = eigs.compute(sa.SortRule.SmallestAlge, maxit=100, tol=1e-9)
        assert nconv == k and eigs.info() == sa.CompInfo.Successful
        ev, X = eigs.eigenvalues(), eigs.eigenvectors()
        assert np.abs(B @ X - X * ev).max() < 1e-8
        out[mode] = (ev, eigs.num_iterations())
    assert np.abs(out["none"][0] - out["rcm"][0]).max() < 1e-8
    assert abs(out["none"][1] - out["rcm"][1]) <= 2, out          # same preconditioner => same convergence history
    # regular inverse: B^{-1} x by CG with the Jacobi preconditioner, reordering forced through the environment
    M = B.tocsc()                                                  # diagonal >= 1, off-diagonal row sums < 0.6: positive definite
    x = np.random.default_rng(4).uniform(-1, 1, n)
    its = {}
    for mode in ("none", "rcm"):
        monkeypatch.setenv("MISPEC_REORDER", mode)
        R = sa.SparseRegularInverse(sp.tril(M).tocsc(), ctx=ctx)
        y = R.solve(x)
        assert np.abs(M @ y - x).max() <= 1e-10 * np.abs(x).max() * 100
        its[mode] = R.last_iterations()
    monkeypatch.delenv("MISPEC_REORDER")
    assert abs(its["none"] - its["rcm"]) <= 1, its
