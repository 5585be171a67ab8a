"""The CPU oracle's IRLM driver against the reference's solver-level tests and known answers (no GPU)."""
import os
import zlib

import numpy as np
import pytest
import scipy.sparse.linalg as spla

import oracle as O
from helpers import EXAMPLE2, RULES_SYM, SPARSE_CASES, cycle_laplacian, sparse_fixture, wanted_by_rule

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "symeigs_golden.npz"))


def run(op, A, k, m, rule, **kw):
    s = O.SymEigsSolver(op, k, m)
    s.init(kw.pop("v0", None))
    nconv = s.compute(getattr(O, rule), **kw)
    ev, U = s.eigenvalues(), s.eigenvectors()
    resid = np.abs(A @ U - U * ev).max() if U.shape[1] else 0.0
    return s, nconv, ev, U, resid


def test_doc_example_diag_1_to_10():
    # SymEigsSolver.h:99-126 / README.md:190-216: M = diag(1..10), nev=3, ncv=6, LargestAlge -> (10, 9, 8)
    s = O.SymEigsSolver(O.Op.diag(np.arange(1.0, 11.0)), 3, 6)
    s.init()
    assert s.compute(O.LargestAlge) == 3 and s.info() == O.Successful
    assert np.allclose(s.eigenvalues(), [10.0, 9.0, 8.0], atol=1e-10)


def test_fixture_is_the_committed_one():
    for n, prob, _, _ in SPARSE_CASES:
        r, c, v = O.gen_sparse_data(n, prob)
        crc = [zlib.crc32(r.tobytes()), zlib.crc32(c.tobytes()), zlib.crc32(v.tobytes())]
        assert crc == list(GOLD[f"crc_{n}"])


@pytest.mark.parametrize("n,prob,k,m", SPARSE_CASES)
@pytest.mark.parametrize("rule", RULES_SYM)
def test_sparse_fixtures_all_rules(n, prob, k, m, rule):
    # test/SymEigs.cpp:133-167 x the 5 selection rules (:78-97): info == Successful, ||AU - UD||_inf <= 1e-9
    A, S = sparse_fixture(n, prob)
    op = O.Op.csc_sym(n, A.indptr, A.indices, A.data, True)
    s, nconv, ev, U, resid = run(op, S, k, m, rule)
    assert s.info() == O.Successful and nconv == k
    assert resid < 1e-9
    # known answer: the wanted part of the dense spectrum (numpy eigvalsh, committed golden)
    want = wanted_by_rule(GOLD[f"spectrum_{n}"], rule, k)
    assert np.abs(np.sort(ev) - want).max() < 1e-9
    # regression pin of the restatement itself
    assert [nconv, s.info(), s.num_iterations(), s.num_operations()] == list(GOLD[f"oracle_{n}_{rule}"])
    assert np.array_equal(ev, GOLD[f"oracle_evals_{n}_{rule}"])
    # default sorting is LargestAlge: descending
    assert np.all(np.diff(ev) <= 0)


@pytest.mark.parametrize("k,m", [(3, 6), (5, 12), (6, 12)])
def test_example1_cycle_laplacian(k, m):
    # test/Example1.cpp:34-68: n = 20, tol 1e-15, sorting SmallestAlge; residual and |lambda - true| <= 1e-9.
    # The eigenvalues are 1 - cos(2 pi j / 20) analytically.
    M = cycle_laplacian(20)
    true = np.sort(1.0 - np.cos(2 * np.pi * np.arange(20) / 20))
    s, nconv, ev, U, resid = run(O.Op.dense_sym(M), M, k, m, "LargestMagn", maxit=1000, tol=1e-15, sorting=O.SmallestAlge)
    assert s.info() == O.Successful and resid < 1e-9
    assert np.abs(true[-k:] - ev).max() < 1e-9


@pytest.mark.parametrize("case", range(3))
def test_example2_near_rank_one(case):
    # test/Example2.cpp:18-50: nev=1, ncv=3, LargestAlge; |lambda - true| <= 1e-8
    M = EXAMPLE2[case]
    s, nconv, ev, U, resid = run(O.Op.dense_sym(M), M, 1, 3, "LargestAlge")
    assert s.info() == O.Successful and resid < 1e-8
    assert abs(ev[0] - np.linalg.eigvalsh(M)[-1]) < 1e-8


def test_example4_zero_matrix_and_null_space_start():
    # test/Example4.cpp:59-92: (1) A = 0; (2) A has a zero eigenvalue and v0 is its eigenvector
    rng = np.random.default_rng(123)
    n = 100
    A = np.zeros((n, n))
    s, nconv, ev, U, resid = run(O.Op.dense_sym(A), A, 3, 6, "LargestAlge", v0=rng.uniform(-1, 1, n), sorting=O.SmallestAlge)
    assert s.info() == O.Successful and resid < 1e-8 and np.abs(ev).max() < 1e-8
    Mm = rng.uniform(-1, 1, (n, n))
    Mm = Mm + Mm.T
    w, V = np.linalg.eigh(Mm)
    w[-1] = 0.0
    A = (V * w) @ V.T
    A = (A + A.T) / 2
    true = np.linalg.eigvalsh(A)
    s, nconv, ev, U, resid = run(O.Op.dense_sym(A), A, 3, 6, "LargestAlge", v0=V[:, -1].copy(), sorting=O.SmallestAlge)
    assert s.info() == O.Successful and resid < 1e-8
    assert np.abs(true[-3:] - ev).max() < 1e-8


def test_constructor_checks_and_zero_start_vector():
    op = O.Op.diag(np.arange(1.0, 11.0))
    for nev, ncv in [(0, 5), (10, 11), (3, 3), (3, 11)]:  # HermEigsBase.h:267-271
        with pytest.raises(ValueError):
            O.SymEigsSolver(op, nev, ncv)
    s = O.SymEigsSolver(op, 3, 6)
    with pytest.raises(ValueError):  # Arnoldi.h:146-148
        s.init(np.zeros(10))


def test_not_converging_is_reported():
    A, S = sparse_fixture(100, 0.1)
    s = O.SymEigsSolver(O.Op.csc_sym(100, A.indptr, A.indices, A.data, True), 10, 20)
    s.init()
    nconv = s.compute(O.SmallestMagn, 2)  # maxit = 2
    assert s.info() == O.NotConverging and nconv < 10 and len(s.eigenvalues()) == nconv


def test_cross_check_with_arpack():
    # scipy eigsh is ARPACK (the algorithm Spectra re-designs): eigenvalue agreement only
    rp, ci, v = O.synth_band_csr(20000, offsets=(1, 2, 3, 100, 101, 1000, 1001))
    import scipy.sparse as sp
    A = sp.csr_matrix((v, ci, rp), shape=(20000, 20000))
    s = O.SymEigsSolver(O.Op.csr(20000, 20000, rp, ci, v), 6, 24)
    s.init()
    assert s.compute(O.LargestAlge, 1000, 1e-11) == 6
    ref = np.sort(spla.eigsh(A, k=6, which="LA", tol=1e-12, ncv=24)[0])[::-1]
    assert np.abs(s.eigenvalues() - ref).max() < 1e-9
    U = s.eigenvectors()
    assert (np.linalg.norm(A @ U - U * s.eigenvalues(), axis=0) / np.linalg.norm(U, axis=0)).max() < 1e-10


def test_shift_invert_back_transform():
    # SymEigsShiftSolver.h:163-169: nu -> 1/nu + sigma before sorting; operator = (A - sigma I)^-1 (dense here)
    M = cycle_laplacian(20)
    sigma = -1e-6  # test/Example1.cpp:72
    inv = np.linalg.inv(M - sigma * np.eye(20))
    inv = (inv + inv.T) / 2
    s = O.SymEigsSolver(O.Op.dense_sym(inv), 3, 6, sigma=sigma)
    s.init()
    assert s.compute(O.LargestMagn, 1000, 1e-15, O.SmallestAlge) == 3
    true = np.sort(1.0 - np.cos(2 * np.pi * np.arange(20) / 20))
    assert np.abs(true[:3] - s.eigenvalues()).max() < 1e-9
