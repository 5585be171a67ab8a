"""The Davidson oracle (oracle/davidson.py) against the reference's own test bar (test/DavidsonSymEigs.cpp:69-123) on
its reproducible sparse fixture, and against LAPACK."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle as O
from oracle import davidson as OD


def davidson_sparse_fixture(n):
    """gen_sym_data_sparse(n) (test/DavidsonSymEigs.cpp:46-67): CSC input (not symmetric) and the symmetric matrix its
    lower triangle defines (what SparseSymMatProd<double> applies)."""
    r, c, v = O.gen_davidson_sparse(n)
    A = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsc()
    A.sort_indices()
    S = (sp.tril(A) + sp.tril(A, -1).T).tocsr()
    return A, S


def test_fixture_generator_matches_the_reference_recipe():
    A, S = davidson_sparse_fixture(50)
    assert np.array_equal(A.diagonal(), np.arange(1, 51, dtype=float))  # coeffRef(i, i) = i + 1 overrides the random entry
    off = A - sp.diags(A.diagonal())
    assert abs(off).max() <= 0.05 and 0.3 < off.nnz / (50 * 49) < 0.7   # 0.1 * (u - 0.5) with probability 0.5
    assert abs(S - S.T).max() == 0


@pytest.mark.parametrize("rule", ["LargestAlge", "SmallestAlge"])
def test_oracle_meets_the_reference_test_bar(rule):
    n, k = 1000, 10
    _, S = davidson_sparse_fixture(n)
    eigs = OD.DavidsonSymEigsSolver(S, k)
    nconv = eigs.compute(getattr(OD, rule))
    assert nconv == k and eigs.info() == OD.Successful            # test/DavidsonSymEigs.cpp:77-80
    evals, evecs = eigs.eigenvalues(), eigs.eigenvectors()
    assert np.abs(S @ evecs - evecs * evals).max() < 1e-10         # :85-89 (100 * dummy_precision)
    full = np.linalg.eigvalsh(S.toarray())
    want = full[::-1][:k] if rule == "LargestAlge" else full[:k]
    assert np.abs(evals - want).max() < 1e-9
    assert 1 <= eigs.num_iterations() < 100


def test_oracle_on_a_dense_matrix_and_with_a_guess():
    # gen_sym_data_dense's recipe (test/DavidsonSymEigs.cpp:33-43) with a seeded numpy matrix
    n, k = 300, 6
    rng = np.random.default_rng(0)
    M = 0.03 * rng.uniform(-1, 1, (n, n))
    A = M + M.T + np.diag(np.arange(1, n + 1, dtype=float))
    eigs = OD.DavidsonSymEigsSolver(A, k)
    assert eigs.compute(OD.LargestAlge) == k and eigs.info() == OD.Successful
    assert np.abs(eigs.eigenvalues() - np.linalg.eigvalsh(A)[::-1][:k]).max() < 1e-9
    # compute_with_guess from a perturbed exact basis
    w, U = np.linalg.eigh(A)
    guess, _ = np.linalg.qr(U[:, -12:] + 1e-3 * rng.uniform(-1, 1, (n, 12)))
    e2 = OD.DavidsonSymEigsSolver(A, k)
    assert e2.compute_with_guess(guess, OD.LargestAlge) == k
    assert e2.info() == OD.Successful
    assert np.abs(e2.eigenvalues() - w[::-1][:k]).max() < 1e-9


def test_size_rules_and_errors():
    A = np.diag(np.arange(1.0, 21.0))
    with pytest.raises(ValueError):
        OD.DavidsonSymEigsSolver(A, 20)  # nev <= n - 1
    s = OD.DavidsonSymEigsSolver(A, 3)  # defaults 2 nev / 10 nev; 10 nev = 30 >= n -> n
    assert (s.init_size, s.max_size, s.corr_size) == (6, 20, 3)
    t = OD.DavidsonSymEigsSolver(A, 8)  # n < init + corr -> n / 3 each
    assert (t.init_size, t.corr_size) == (6, 6)
