"""The CPU oracle against the reference's own L2 unit tests and known answers (no GPU).

Each block names the reference test it re-creates (paths relative to /root/reference/).  Eigen is
not available, so Eigen-generated random inputs are replaced by numpy ones of the same shape/size.
"""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle as O
from helpers import random_tridiag, sparse_fixture


def test_simple_random_matches_minstd_and_known_states():
    # first states 16807, 282475249, 1622650073; libstdc++ minstd_rand0 is the same LCG
    assert list(O.lcg_states(1, 3)) == [16807, 282475249, 1622650073]
    for seed in (1, 2, 12345, 2147483646, 16807):
        assert np.array_equal(O.lcg_states(seed, 2000), O.minstd_states(seed, 2000))
    v = O.simple_random(5, seed=0)  # seed 0 -> state 1 (SimpleRandom.h:92-96)
    assert np.allclose(v, np.array([16807, 282475249, 1622650073, 984943658, 1144108930]) / 2147483647.0 - 0.5, rtol=0, atol=0)
    assert np.all(np.abs(O.simple_random(10000, 7)) <= 0.5)


def test_givens_rotation_properties():
    # test/Givens.cpp:64-99 — 100000 pairs, 10% zeros, U(-100,100); c*x - s*y = r, s*x + c*y = 0 to 1e-12
    rng = np.random.default_rng(0)
    x = rng.uniform(-100, 100, 100000)
    y = rng.uniform(-100, 100, 100000)
    x[rng.uniform(size=x.size) < 0.1] = 0.0
    y[rng.uniform(size=y.size) < 0.1] = 0.0
    for xi, yi in zip(x[:20000], y[:20000]):
        r, c, s = O.givens(xi, yi)
        assert abs(c * xi - s * yi - r) < 1e-12 * max(1.0, abs(r))
        assert abs(s * xi + c * yi) < 1e-12 * max(1.0, abs(r))
        assert abs(c * c + s * s - 1.0) < 1e-14
        assert r >= 0
    # the Taylor branch (t < 0.1 eps^(1/4)) and the conventions c = x/r, s = -y/r
    for xi, yi in [(1.0, 1e-7), (-3.0, 2e-6), (1e-9, 5.0), (0.0, -2.0), (4.0, 0.0), (0.0, 0.0), (-1.0, 0.0)]:
        r, c, s = O.givens(xi, yi)
        assert r == pytest.approx(np.hypot(xi, yi), rel=1e-15)
        if r > 0:
            assert c == pytest.approx(xi / r, abs=1e-15) and s == pytest.approx(-yi / r, abs=1e-15)
        else:
            assert (c, s) == (1.0, 0.0)


def test_eigen_make_givens():
    rng = np.random.default_rng(1)
    for p, q in rng.uniform(-3, 3, (2000, 2)):
        c, s = O.eigen_make_givens(p, q)
        # G' [p; q] = [r; 0] with G = [c s; -s c] (Eigen convention): s*p + c*q = 0
        assert abs(s * p + c * q) < 1e-14 * np.hypot(p, q) + 1e-300
        assert abs(c * c + s * s - 1) < 1e-14
    assert O.eigen_make_givens(2.0, 0.0) == (1.0, 0.0)
    assert O.eigen_make_givens(-2.0, 0.0) == (-1.0, 0.0)
    assert O.eigen_make_givens(0.0, 2.0) == (0.0, -1.0)


@pytest.mark.parametrize("shift", [1.2345, 0.6789, 0.0])
def test_tridiag_qr_identities(shift):
    # test/QR.cpp "QR of real tridiagonal matrix", n = 100, tol 1e-12 (:20-99 run_test)
    n = 100
    T = random_tridiag(n, 123)
    R, QtHQ, Q = O.tridiag_qr(T, shift)
    I = np.eye(n)
    assert np.abs(Q.T @ Q - I).max() < 1e-12
    assert np.abs(Q @ Q.T - I).max() < 1e-12
    assert np.abs(np.tril(R, -1)).max() == 0.0
    assert np.abs(T - shift * I - Q @ R).max() < 1e-12
    assert np.abs(QtHQ - Q.T @ T @ Q).max() < 1e-12


def test_tridiag_qr_deflates_tiny_subdiagonals():
    # UpperHessenbergQR.h:533-539 / :676-682
    T = np.diag([1.0, 2.0, 3.0]) + np.diag([1e-20, 0.5], -1) + np.diag([1e-20, 0.5], 1)
    _, QtHQ, _ = O.tridiag_qr(T, 0.3)
    assert QtHQ[1, 0] == 0.0 and QtHQ[0, 1] == 0.0


@pytest.mark.parametrize("n", [2, 3, 10, 100])
def test_tridiag_eigen_residual(n):
    # test/Eigen.cpp:67-110 "Eigen decomposition of symmetric real tridiagonal matrix": ||HU - UD||_inf < 1e-12
    T = random_tridiag(n, 321 + n)
    ev, U = O.tridiag_eigen(T)
    assert np.abs(T @ U - U * ev).max() < 1e-12
    assert np.abs(U.T @ U - np.eye(n)).max() < 1e-12
    assert np.allclose(np.sort(ev), np.linalg.eigvalsh(T), atol=1e-12)


def test_tridiag_eigen_zero_and_diagonal():
    ev, U = O.tridiag_eigen(np.zeros((5, 5)))  # TridiagEigen.h:142-150 early exit
    assert np.all(ev == 0) and np.array_equal(U, np.eye(5))
    ev, U = O.tridiag_eigen(np.diag([3.0, -1.0, 2.0]))
    assert np.array_equal(ev, [3.0, -1.0, 2.0]) and np.array_equal(U, np.eye(3))


def test_argsort_rules():
    v = np.array([0.5, -3.0, 2.0, -0.1, 1.5, -2.5])
    assert list(v[O.argsort(O.LargestMagn, v)]) == [-3.0, -2.5, 2.0, 1.5, 0.5, -0.1]
    assert list(v[O.argsort(O.LargestAlge, v)]) == [2.0, 1.5, 0.5, -0.1, -2.5, -3.0]
    assert list(v[O.argsort(O.SmallestMagn, v)]) == [-0.1, 0.5, 1.5, 2.0, -2.5, -3.0]
    assert list(v[O.argsort(O.SmallestAlge, v)]) == [-3.0, -2.5, -0.1, 0.5, 1.5, 2.0]
    # BothEnds: largest, smallest, 2nd largest, 2nd smallest, ... (SelectionRule.h:265-284)
    assert list(v[O.argsort(O.BothEnds, v)]) == [2.0, -3.0, 1.5, -2.5, 0.5, -0.1]
    with pytest.raises(ValueError):
        O.argsort(O.LargestReal, v)


def test_sparse_operators_against_scipy():
    # test/SparseSymMatProd.cpp:37-54 / SparseGenMatProd.cpp:37-53: op * x == mat * x
    A, S = sparse_fixture(100, 0.1)
    x = np.random.default_rng(5).standard_normal(100)
    y = O.Op.csc_sym(100, A.indptr, A.indices, A.data, True).perform_op(x)
    assert np.allclose(y, S @ x, rtol=0, atol=1e-13)
    Su = (sp.triu(A) + sp.triu(A, 1).T).tocsr()
    yu = O.Op.csc_sym(100, A.indptr, A.indices, A.data, False).perform_op(x)
    assert np.allclose(yu, Su @ x, rtol=0, atol=1e-13)
    Ar = A.tocsr()
    assert np.allclose(O.Op.csr(100, 100, Ar.indptr, Ar.indices, Ar.data).perform_op(x), A @ x, atol=1e-13)
    assert np.allclose(O.Op.csc(100, 100, A.indptr, A.indices, A.data).perform_op(x), A @ x, atol=1e-13)


def test_gen_sparse_data_fixture_shape():
    # the libstdc++-only fixture: density close to prob, values in [-0.5, 0.5), row-major insertion order
    r, c, v = O.gen_sparse_data(1000, 0.01)
    assert 9000 < len(v) < 11000 and np.all(np.abs(v) <= 0.5)
    assert np.all(np.diff(r.astype(np.int64) * 1000 + c) > 0)


def test_synth_band_matrix():
    rp, ci, v = O.synth_band_csr(5000, offsets=(1, 2, 3, 100, 101))
    A = sp.csr_matrix((v, ci, rp), shape=(5000, 5000))
    assert abs(A - A.T).max() == 0.0
    assert np.all(np.diff(rp)[200:-200] == 11) and np.all(np.abs(v) <= 0.5)
    rp2, ci2, v2 = O.synth_band_csr(5000, offsets=(1, 2, 3, 100, 101), symmetric=False)
    assert np.array_equal(ci, ci2) and abs(sp.csr_matrix((v2, ci2, rp2)) - sp.csr_matrix((v2, ci2, rp2)).T).max() > 0.1


@pytest.mark.parametrize("symmetric", [True, False])
def test_factorization_identities(symmetric):
    # test/Arnoldi.cpp:19-85: n = 10, m = 6; init -> k = 1; factorize to m/2, to m;
    # A V - V H = f e' column-wise and V'V = I, tol 1e-12
    n, m = 10, 6
    rng = np.random.default_rng(123)
    M = rng.uniform(-1, 1, (n, n))
    A = M + M.T if symmetric else M
    op = O.Op.dense_sym(A) if symmetric else O.Op.dense_gen(A)
    fac = O.Factorization(op, m, symmetric)
    fac.init(rng.uniform(-1, 1, n))
    assert fac.subspace_dim() == 1

    def check(k):
        V, H, f = fac.matrices()
        V, H = V[:, :k], H[:k, :k]
        resid = A @ V - V @ H
        if k > 1:
            assert np.abs(resid[:, :k - 1]).max() < 1e-12
        assert np.abs(resid[:, -1] - f).max() < 1e-12
        return V

    check(1)
    fac.factorize_from(1, m // 2)
    assert fac.subspace_dim() == m // 2
    check(m // 2)
    fac.factorize_from(m // 2, m)
    assert fac.subspace_dim() == m
    V = check(m)
    assert np.abs(V.T @ V - np.eye(m)).max() < 1e-12
    assert fac.num_operations() == 2 + (m - 1)  # Appendix A: 2 in init + one per step
    with pytest.raises(ValueError):
        O.Factorization(op, m, symmetric).factorize_from(3, 5)  # from_k > current dimension (Lanczos.h:70-75)
