"""Generalized symmetric eigen solver in regular-inverse mode on the GPU (SymGEigsSolver.h:224-238): the operator
B^{-1}A with a device conjugate gradient and B-inner products inside the device Lanczos factorisation.
Parity: the reference's own fixtures and bar (test/SymGEigsRegInv.cpp:18-145: ||AU - BUD||_inf <= 1e-9, info ==
Successful, compute(selection, 100)) and the oracle's restatement (eigenvalues to 1e-9, same nconv)."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle as O
import spectra_amd as sa

pytestmark = pytest.mark.gpu

GEIGS_CASES = [(10, 0.5, 3, 6), (100, 0.1, 10, 20), (1000, 0.01, 20, 50)]  # test/SymGEigsRegInv.cpp:109-145
RULES = ["LargestMagn", "LargestAlge", "SmallestAlge", "BothEnds"]        # SmallestMagn is allow_fail upstream


def geigs_fixture(n, prob):
    r, c, v = O.gen_sparse_data(n, prob)
    A = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsc()
    B = (A.T @ A + 0.1 * sp.identity(n)).tocsc()
    As = (sp.tril(A) + sp.tril(A, -1).T).tocsc()
    return A, B, As


def test_regular_inverse_operator(ctx):
    A, B, _ = geigs_fixture(100, 0.1)
    Bop = sa.SparseRegularInverse(B, ctx=ctx)
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, 100)
    assert np.abs(Bop.perform_op(x) - B @ x).max() <= 1e-13
    y = Bop.solve(x)
    assert 0 < Bop.last_iterations() <= 200
    assert np.linalg.norm(B @ y - x) <= 1e-12 * np.linalg.norm(x)
    oracle = O.SymGEigsRegInvSolver(A, B, 5, 12)
    y0, it0 = oracle.cg_solve(x)
    assert np.abs(y - y0).max() <= 1e-11 * np.abs(y0).max() and abs(Bop.last_iterations() - it0) <= 5
    assert not Bop.solve(np.zeros(100)).any() and Bop.last_iterations() == 0
    with pytest.raises(ValueError, match="square"):
        sa.SparseRegularInverse(sp.random(5, 6, density=0.5, format="csc"), ctx=ctx)


@pytest.mark.parametrize("n,prob,k,m", GEIGS_CASES)
@pytest.mark.parametrize("rule", RULES)
def test_reginv_fixtures(ctx, n, prob, k, m, rule):
    A, B, As = geigs_fixture(n, prob)
    eigs = sa.SymGEigsSolver(sa.SparseSymMatProd(A, ctx=ctx), sa.SparseRegularInverse(B, ctx=ctx), k, m)
    eigs.init()
    nconv = eigs.compute(sa.SortRule[rule], 100)
    assert eigs.info() == sa.CompInfo.Successful and nconv == k
    ev, U = eigs.eigenvalues(), eigs.eigenvectors()
    assert np.abs(As @ U - (B @ U) * ev).max() <= 1e-9          # the reference's bar
    assert np.abs(U.T @ (B @ U) - np.eye(k)).max() <= 1e-9       # B-orthonormal
    oe = O.SymGEigsRegInvSolver(A, B, k, m)
    oe.init()
    assert oe.compute(getattr(O, rule), 100) == k
    assert np.abs(ev - oe.eigenvalues()).max() <= 1e-9 * max(1.0, np.abs(ev).max())
    assert eigs.num_operations() == pytest.approx(oe.num_operations(), rel=0.2)


def test_generalized_at_scale(ctx):
    # 200k x 200k: A the banded benchmark pattern, B a consistent-mass-like tridiagonal SPD matrix
    n, k, m = 200_000, 6, 20
    rp, ci, v = O.synth_band_csr(n, offsets=(1, 2, 3, 500, 501))
    A = sp.csr_matrix((v, ci, rp), shape=(n, n))
    B = sp.diags([np.full(n - 1, 1.0 / 6.0), np.full(n, 4.0 / 6.0), np.full(n - 1, 1.0 / 6.0)], [-1, 0, 1], format="csc")
    eigs = sa.SymGEigsSolver(sa.SparseSymMatProd(sp.tril(A).tocsc(), ctx=ctx), sa.SparseRegularInverse(B, ctx=ctx), k, m)
    eigs.init()
    assert eigs.compute(sa.SortRule.LargestAlge, 300, 1e-10) == k
    ev, U = eigs.eigenvalues(), eigs.eigenvectors()
    R = A @ U - (B @ U) * ev
    assert (np.linalg.norm(R, axis=0) / np.linalg.norm(B @ U, axis=0)).max() <= 1e-8
    assert eigs.residuals().max() <= 1e-8
    assert np.abs(U.T @ (B @ U) - np.eye(k)).max() <= 1e-9


# ---- shift modes (SymGEigsShiftSolver.h; test/SymGEigsShift.cpp sparse-sparse cases, sigma = 1.2345, k = 10, m = 20) ----
def shift_fixture(mode):
    A, B, As = geigs_fixture(100, 0.1)
    if mode == "Buckling":  # gen_sparse_data(100, KG, K): the pencil is (K, KG) and the inner product is K
        return B, A, (sp.tril(B) + sp.tril(B, -1).T).tocsc(), As
    return A, B, As, B


@pytest.mark.parametrize("mode", ["ShiftInvert", "Buckling", "Cayley"])
@pytest.mark.parametrize("rule", RULES)
def test_shift_modes_fixtures(ctx, mode, rule):
    A, B, As, Bs = shift_fixture(mode)
    k, m, sigma = 10, 20, 1.2345
    op = sa.SymShiftInvert(A, B, ctx=ctx)
    Bop = sa.SparseSymMatProd(A if mode == "Buckling" else B, ctx=ctx)
    eigs = sa.SymGEigsShiftSolver(op, Bop, k, m, sigma, mode)
    eigs.init()
    nconv = eigs.compute(sa.SortRule[rule], 100)
    assert eigs.info() == sa.CompInfo.Successful and nconv == k
    ev, U = eigs.eigenvalues(), eigs.eigenvectors()
    assert np.abs(As @ U - (Bs @ U) * ev).max() <= 1e-9          # the reference's bar
    oe = O.SymGEigsShiftSolver(A, B, k, m, sigma, mode)
    oe.init()
    assert oe.compute(getattr(O, rule), 100) == k
    assert np.abs(ev - oe.eigenvalues()).max() <= 1e-8 * max(1.0, np.abs(ev).max())


def test_pencil_operator_and_errors(ctx):
    A, B, As, Bs = shift_fixture("ShiftInvert")
    op = sa.SymShiftInvert(A, B, ctx=ctx)
    op.set_shift(0.5)
    x = np.random.default_rng(2).uniform(-1, 1, 100)
    ref = np.linalg.solve((As - 0.5 * B).toarray(), x)
    assert np.abs(op.perform_op(x) - ref).max() <= 1e-10 * np.abs(ref).max()
    with pytest.raises(ValueError, match="same size"):
        sa.SymShiftInvert(A, sp.identity(50, format="csc"), ctx=ctx)
    with pytest.raises(ValueError, match="sigma cannot be zero"):
        sa.SymGEigsShiftSolver(op, sa.SparseSymMatProd(B, ctx=ctx), 5, 12, 0.0, "Cayley")


def test_banded_pencil_at_scale(ctx):
    # 1D stiffness / mass pencil (tridiagonal K and M, n = 400k): the six eigenvalues closest to sigma
    n, k, m, sigma = 400_000, 6, 20, 0.0
    K = sp.diags([np.full(n - 1, -1.0), np.full(n, 2.0), np.full(n - 1, -1.0)], [-1, 0, 1], format="csc")
    M = sp.diags([np.full(n - 1, 1.0 / 6.0), np.full(n, 4.0 / 6.0), np.full(n - 1, 1.0 / 6.0)], [-1, 0, 1], format="csc")
    op = sa.SymShiftInvert(K, M, ctx=ctx)
    eigs = sa.SymGEigsShiftSolver(op, sa.SparseSymMatProd(M, ctx=ctx), k, m, sigma, "ShiftInvert")
    eigs.init()
    assert eigs.compute(sa.SortRule.LargestMagn, 300, 1e-10) == k
    ev, U = np.sort(eigs.eigenvalues()), eigs.eigenvectors()
    j = np.arange(1, k + 1)
    theta = j * np.pi / (n + 1)
    exact = 12.0 * np.sin(theta / 2) ** 2 / (2.0 + np.cos(theta))   # (2 - 2 cos t) / ((4 + 2 cos t) / 6), cancellation-free
    # K has condition number ~ n^2 = 1.6e11: the smallest eigenvalues carry a relative error of that times epsilon
    assert np.abs(ev / exact - 1.0).max() <= 1e-4
    R = K @ U - (M @ U) * eigs.eigenvalues()
    assert (np.linalg.norm(R, axis=0) / np.linalg.norm(M @ U, axis=0)).max() <= 1e-8


# ---- Cholesky mode (SymGEigsSolver.h:142-208; test/SymGEigsCholesky.cpp sparse cases, test/Example3.cpp) --------------
def test_sparse_cholesky_operator(ctx):
    A, B, _ = geigs_fixture(100, 0.1)
    Bop = sa.SparseCholesky(B, ctx=ctx)
    assert Bop.info() == sa.CompInfo.Successful
    L = np.linalg.cholesky(B.toarray())
    x = np.random.default_rng(4).uniform(-1, 1, 100)
    assert np.abs(Bop.lower_triangular_solve(x) - np.linalg.solve(L, x)).max() <= 1e-11
    assert np.abs(Bop.upper_triangular_solve(x) - np.linalg.solve(L.T, x)).max() <= 1e-11
    bad = sa.SparseCholesky(sp.diags([1.0, -1.0, 2.0], format="csc"), ctx=ctx)      # not positive definite
    assert bad.info() == sa.CompInfo.NumericalIssue
    with pytest.raises(ValueError, match="positive definite"):
        sa.SymGEigsSolver(sa.SparseSymMatProd(sp.identity(3, format="csc"), ctx=ctx), bad, 1, 2, "Cholesky")
    with pytest.raises(ValueError, match="banded"):          # large and not banded: regular-inverse mode is the way
        sa.SparseCholesky(sp.random(5000, 5000, density=2e-3, random_state=0, format="csc") + 10 * sp.identity(5000, format="csc"), ctx=ctx)


def banded_pd(n, b, seed):
    rng = np.random.default_rng(seed)
    diags = [rng.uniform(-0.5, 0.5, n - d) for d in range(1, b + 1)]
    return sp.diags([rng.uniform(0.0, 1.0, n) + b + 0.5] + diags + diags, [0] + list(range(1, b + 1)) + [-d for d in range(1, b + 1)], format="csc")


# (half-bandwidths above 8: the Cholesky halves of the wave-per-chunk solve, k_chunk_solve_wave modes 0 + u_out and 2 — round 6)
@pytest.mark.parametrize("n,b", [(5000, 1), (20_000, 3), (300_001, 2), (1_000_000, 4), (60_007, 12), (100_000, 32), (70_000, 64)])
def test_banded_cholesky_beyond_the_dense_limit(ctx, n, b):
    # SparseCholesky for n > 4096: a banded B is factored by the partitioned band factorisation;
    # lower_triangular_solve / upper_triangular_solve are G^{-1} x and G^{-T} x of a factor with G G' = B (any such factor
    # gives the same generalized eigenpairs — the reference's own L is that of a fill-reducing permutation)
    import scipy.sparse.linalg as spla

    B = banded_pd(n, b, seed=n)
    Bop = sa.SparseCholesky(sp.tril(B).tocsc(), ctx=ctx)
    assert Bop.info() == sa.CompInfo.Successful
    x = np.random.default_rng(1).uniform(-1, 1, n)
    u = Bop.lower_triangular_solve(x)                 # G^{-1} x
    z = Bop.upper_triangular_solve(u)                 # G^{-T} G^{-1} x = B^{-1} x
    ref = spla.splu(B).solve(x)
    assert np.abs(z - ref).max() <= 1e-11 * np.abs(ref).max()
    # G^{-1} B G^{-T} = I  on a probe vector, and |G^{-1} x|^2 = x' B^{-1} x
    w = Bop.lower_triangular_solve(B @ Bop.upper_triangular_solve(x))
    assert np.abs(w - x).max() <= 1e-11
    assert abs(u @ u - x @ ref) <= 1e-11 * abs(x @ ref)
    notpd = B - (b + 1.0) * sp.identity(n, format="csc")
    assert sa.SparseCholesky(sp.tril(notpd).tocsc(), ctx=ctx).info() == sa.CompInfo.NumericalIssue


def test_banded_cholesky_of_a_scrambled_matrix(ctx):
    # round 5: a banded B in a scattering row order is reordered at construction (reverse Cuthill-McKee); the factor the solver
    # works with is then P'G with G G' = P B P' — (P'G)^{-1} x = G^{-1} P x, (P'G)^{-T} x = P' G^{-T} x: the same identities hold
    import scipy.sparse.linalg as spla

    n, b = 30_000, 4
    q = np.random.default_rng(3).permutation(n)
    B = banded_pd(n, b, seed=5)[q][:, q].tocsc()
    Bop = sa.SparseCholesky(sp.tril(B).tocsc(), ctx=ctx)
    assert Bop.info() == sa.CompInfo.Successful
    x = np.random.default_rng(1).uniform(-1, 1, n)
    u = Bop.lower_triangular_solve(x)
    z = Bop.upper_triangular_solve(u)
    ref = spla.splu(B).solve(x)
    assert np.abs(z - ref).max() <= 1e-11 * np.abs(ref).max()
    w = Bop.lower_triangular_solve(B @ Bop.upper_triangular_solve(x))
    assert np.abs(w - x).max() <= 1e-11 and abs(u @ u - x @ ref) <= 1e-11 * abs(x @ ref)


def test_cholesky_mode_on_a_large_banded_pencil(ctx):
    # SymGEigsSolver<SparseSymMatProd, SparseCholesky, GEigsMode::Cholesky> at n = 200 000 (was limited to 4096): against
    # the regular-inverse mode of the same pencil (another algorithm: B^{-1} A with B-inner products) and scipy
    import scipy.sparse.linalg as spla

    n = 200_000
    rp, ci, v = O.synth_band_csr(n, offsets=(1, 2, 3, 50, 51))
    A = sp.csr_matrix((v, ci, rp), shape=(n, n))
    B = sp.diags([np.full(n - 1, 1.0 / 6.0), np.full(n, 4.0 / 6.0), np.full(n - 1, 1.0 / 6.0)], [-1, 0, 1], format="csc")
    aop = sa.SparseSymMatProd(sp.tril(A).tocsc(), ctx=ctx)
    eigs = sa.SymGEigsSolver(aop, sa.SparseCholesky(B, ctx=ctx), 6, 24, "Cholesky")
    eigs.init()
    assert eigs.compute(sa.SortRule.LargestAlge, 1000, 1e-10) == 6
    ev, U = eigs.eigenvalues(), eigs.eigenvectors()
    res = np.linalg.norm(A @ U - (B @ U) * ev, axis=0) / np.linalg.norm(B @ U, axis=0)
    assert res.max() <= 1e-8
    assert np.abs(U.T @ (B @ U) - np.eye(6)).max() <= 1e-9            # B-orthonormal (SymGEigsSolver.h:196-207)
    ri = sa.SymGEigsSolver(aop, sa.SparseRegularInverse(B, ctx=ctx), 6, 24)
    ri.init()
    assert ri.compute(sa.SortRule.LargestAlge, 1000, 1e-10) == 6
    assert np.abs(ri.eigenvalues() - ev).max() <= 1e-8


@pytest.mark.parametrize("n,prob,k,m", GEIGS_CASES)
@pytest.mark.parametrize("rule", RULES)
@pytest.mark.parametrize("orth", ["onesweep", "reference"])
def test_cholesky_fixtures(ctx, n, prob, k, m, rule, orth):
    # the Cholesky mode is a STANDARD symmetric problem for L^{-1} A L^{-T}: device-driven steps, one-sweep by default
    A, B, As = geigs_fixture(n, prob)
    eigs = sa.SymGEigsSolver(sa.SparseSymMatProd(A, ctx=ctx), sa.SparseCholesky(B, ctx=ctx), k, m, "Cholesky")
    eigs.set_orth_mode(orth)
    eigs.init()
    nconv = eigs.compute(sa.SortRule[rule], 100)
    assert eigs.info() == sa.CompInfo.Successful and nconv == k
    ev, U = eigs.eigenvalues(), eigs.eigenvectors()
    assert np.abs(As @ U - (B @ U) * ev).max() <= 1e-9          # test/SymGEigsCholesky.cpp:85-89
    assert np.abs(U.T @ (B @ U) - np.eye(k)).max() <= 1e-9
    oe = O.SymGEigsCholeskySolver(A, B, k, m)
    oe.init()
    assert oe.compute(getattr(O, rule), 100) == k
    assert np.abs(ev - oe.eigenvalues()).max() <= 1e-9 * max(1.0, np.abs(ev).max())
    assert eigs.num_operations() == pytest.approx(oe.num_operations(), rel=0.2)
    info = eigs.orth_info()
    assert info["mode"] == orth and (info["lagged_steps"] > 0) == (orth == "onesweep")


def test_example3_issue115(ctx):
    # test/Example3.cpp:61-94 (case 1): A = M positive SEMI-definite, B = C + 1e5 M, nef = 4, ncv = 5
    C_tri = [(0, 0, 1.1807575e+08), (1, 1, 304744.5), (1, 5, -152372.25), (2, 2, 304744.5), (2, 4, 152372.25), (3, 3, 15403.85),
             (4, 2, 152372.25), (4, 4, 101581.5), (5, 1, -152372.25), (5, 5, 101581.5)]
    M_tri = [(0, 0, 1000.0), (1, 1, 1000.0), (2, 2, 1000.0)]
    mk = lambda tri: sp.coo_matrix(([v for _, _, v in tri], ([i for i, _, _ in tri], [j for _, j, _ in tri])), shape=(6, 6)).tocsc()
    A, B = mk(M_tri), (mk(C_tri) + 1.0e5 * mk(M_tri)).tocsc()
    Bop = sa.SparseCholesky(B, ctx=ctx)
    assert Bop.info() == sa.CompInfo.Successful
    eigs = sa.SymGEigsSolver(sa.SparseSymMatProd(A, ctx=ctx), Bop, 4, 5, "Cholesky")
    eigs.init()
    eigs.compute(sa.SortRule.LargestMagn)
    assert eigs.info() == sa.CompInfo.Successful
    ev, U = eigs.eigenvalues(), eigs.eigenvectors()
    assert np.abs(A @ U - (B @ U) * ev).max() <= 1e-9
