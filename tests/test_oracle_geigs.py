"""Generalized symmetric eigen solver, regular-inverse mode, on the oracle — pinned the way test/SymGEigsRegInv.cpp
pins the reference: its reproducible fixtures (sprand(n, prob) with the libstdc++ engine, B = A'A + 0.1 I,
:18-44), `compute(selection, 100)`, info == Successful and ||A U - B U D||_inf <= 1e-9 (:47-83) — plus the dense
generalized eigenvalues of scipy as an independent check, and the conjugate-gradient restatement on its own."""
import numpy as np
import pytest
import scipy.linalg as sla
import scipy.sparse as sp

import oracle as O

GEIGS_CASES = [(10, 0.5, 3, 6), (100, 0.1, 10, 20), (1000, 0.01, 20, 50)]  # test/SymGEigsRegInv.cpp:109-145
RULES = ["LargestMagn", "LargestAlge", "SmallestAlge", "BothEnds"]        # SmallestMagn is allow_fail upstream (:99-102)


def geigs_fixture(n, prob):
    """gen_sparse_data(n, A, B, prob) of test/SymGEigsRegInv.cpp:35-44: A = sprand (lower triangle used), B = A'A + 0.1 I."""
    r, c, v = O.gen_sparse_data(n, prob)
    A = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsc()
    B = (A.T @ A + 0.1 * sp.identity(n)).tocsc()
    As = (sp.tril(A) + sp.tril(A, -1).T).tocsc()
    return A, B, As


def test_conjugate_gradient_restatement():
    A, B, _ = geigs_fixture(100, 0.1)
    s = O.SymGEigsRegInvSolver(A, B, 5, 12)
    rng = np.random.default_rng(0)
    for _ in range(3):
        b = rng.uniform(-1, 1, 100)
        x, it = s.cg_solve(b)
        assert 0 < it <= 200
        assert np.linalg.norm(B @ x - b) <= 1e-12 * np.linalg.norm(b)
    x, it = s.cg_solve(np.zeros(100))
    assert it == 0 and not x.any()
    assert np.abs(s.b_product(b) - B @ b).max() <= 1e-13


@pytest.mark.parametrize("n,prob,k,m", GEIGS_CASES)
@pytest.mark.parametrize("rule", RULES)
def test_reginv_fixtures(n, prob, k, m, rule):
    A, B, As = geigs_fixture(n, prob)
    s = O.SymGEigsRegInvSolver(A, B, k, m)
    s.init()
    nconv = s.compute(getattr(O, rule), 100)
    assert s.info() == O.Successful and nconv == k
    ev, U = s.eigenvalues(), s.eigenvectors()
    assert np.abs(As @ U - (B @ U) * ev).max() <= 1e-9          # the reference's bar
    assert np.abs(U.T @ (B @ U) - np.eye(k)).max() <= 1e-9       # B-orthonormal Ritz vectors
    full = sla.eigh(As.toarray(), B.toarray(), eigvals_only=True)
    from helpers import wanted_by_rule

    want = wanted_by_rule(full, rule, k)
    assert np.abs(np.sort(ev) - np.sort(want)).max() <= 1e-8 * max(1.0, np.abs(want).max())


# ---- shift modes (test/SymGEigsShift.cpp: sparse-sparse cases :121-141, :214-234, :307-327; sigma = 1.2345) ----------
def shift_fixture(mode):
    A, B, As = geigs_fixture(100, 0.1)
    if mode == "Buckling":  # gen_sparse_data(100, KG, K): K = KG'KG + 0.1 I is the first operand, KG the second
        return B, A, (sp.tril(B) + sp.tril(B, -1).T).tocsc(), As
    return A, B, As, B


@pytest.mark.parametrize("mode", ["ShiftInvert", "Buckling", "Cayley"])
@pytest.mark.parametrize("rule", RULES)
def test_shift_modes_fixtures(mode, rule):
    A, B, As, Bs = shift_fixture(mode)
    k, m, sigma = 10, 20, 1.2345
    s = O.SymGEigsShiftSolver(A, B, k, m, sigma, mode)
    s.init()
    nconv = s.compute(getattr(O, rule), 100)
    assert s.info() == O.Successful and nconv == k
    ev, U = s.eigenvalues(), s.eigenvectors()
    assert np.abs(As @ U - (Bs @ U) * ev).max() <= 1e-9          # the reference's bar (:88-92)
    # the pairs found are eigenpairs of the pencil: compare with the dense generalized spectrum
    full = sla.eigvals(As.toarray(), Bs.toarray())
    full = np.sort(full.real[np.abs(full.imag) < 1e-8])
    assert all(np.abs(full - lam).min() <= 1e-7 * max(1.0, abs(lam)) for lam in ev)


# ---- Cholesky mode (test/SymGEigsCholesky.cpp sparse cases :172-207; test/Example3.cpp: issue #115) ------------------
@pytest.mark.parametrize("n,prob,k,m", GEIGS_CASES)
@pytest.mark.parametrize("rule", ["LargestMagn", "LargestAlge", "SmallestMagn", "SmallestAlge", "BothEnds"])
def test_cholesky_fixtures(n, prob, k, m, rule):
    A, B, As = geigs_fixture(n, prob)
    s = O.SymGEigsCholeskySolver(A, B, k, m)
    s.init()
    nconv = s.compute(getattr(O, rule), 100)
    if rule == "SmallestMagn" and s.info() != O.Successful:
        pytest.skip("allow_fail upstream (test/SymGEigsCholesky.cpp:112-115)")
    assert s.info() == O.Successful and nconv == k
    ev, U = s.eigenvalues(), s.eigenvectors()
    assert np.abs(As @ U - (B @ U) * ev).max() <= 1e-9


def example3_case1():
    C_tri = [(0, 0, 1.1807575e+08), (1, 1, 304744.5), (1, 5, -152372.25), (2, 2, 304744.5), (2, 4, 152372.25), (3, 3, 15403.85),
             (4, 2, 152372.25), (4, 4, 101581.5), (5, 1, -152372.25), (5, 5, 101581.5)]
    M_tri = [(0, 0, 1000.0), (1, 1, 1000.0), (2, 2, 1000.0)]
    mk = lambda tri: sp.coo_matrix(([v for _, _, v in tri], ([i for i, _, _ in tri], [j for _, j, _ in tri])), shape=(6, 6)).tocsc()
    return mk(M_tri), mk(C_tri)


def test_example3_issue115_case1():
    # test/Example3.cpp:61-94: A = M (positive semi-definite), B = C + shift M, nef = 4, ncv = 5, LargestMagn
    M, Cm = example3_case1()
    shift = 1.0e5
    A, B = M, (Cm + shift * M).tocsc()
    s = O.SymGEigsCholeskySolver(A, B, 4, 5)
    s.init()
    s.compute(O.LargestMagn)
    assert s.info() == O.Successful
    ev, U = s.eigenvalues(), s.eigenvectors()
    assert np.abs(A @ U - (B @ U) * ev).max() <= 1e-9
