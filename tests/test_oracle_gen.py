"""The CPU oracle's general (non-symmetric) path against the reference's own tests (no GPU):
test/QR.cpp (Hessenberg and double-shift cases), test/Schur.cpp, test/Eigen.cpp (Hessenberg case),
test/GenEigs.cpp (sparse fixtures x 6 selection rules)."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle as O

GEN_CASES = [(10, 0.5, 3, 6), (100, 0.1, 10, 30), (1000, 0.01, 20, 50)]  # test/GenEigs.cpp:143-174
RULES_GEN = ["LargestMagn", "LargestReal", "LargestImag", "SmallestMagn", "SmallestReal", "SmallestImag"]
ALLOW_FAIL = {"SmallestMagn", "SmallestImag"}  # test/GenEigs.cpp:98,106


def hessenberg(n, seed):
    return np.triu(np.random.default_rng(seed).uniform(-1, 1, (n, n)), -1)


def gen_fixture(n, prob):
    r, c, v = O.gen_sparse_data(n, prob)
    A = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsc()
    A.sort_indices()
    return A


@pytest.mark.parametrize("shift", [1.2345, 0.6789])
def test_hessenberg_qr_identities(shift):
    # test/QR.cpp:101-114 "QR of real upper Hessenberg matrix", n = 100, tol 1e-12
    n = 100
    H = hessenberg(n, 123)
    Q, QtHQ = O.hess_qr(H, shift)
    I = np.eye(n)
    assert np.abs(Q.T @ Q - I).max() < 1e-12 and np.abs(Q @ Q.T - I).max() < 1e-12
    assert np.abs(np.tril(Q.T @ (H - shift * I), -1)).max() < 1e-12  # R upper triangular
    assert np.abs(QtHQ - Q.T @ H @ Q).max() < 1e-12


def test_double_shift_qr():
    # test/QR.cpp:136-175: H(1,0) = 0, s = 2, t = 3; Q equals Householder QR of H^2 - sH + tI up to signs
    n = 100
    H = hessenberg(n, 123)
    H[1, 0] = 0.0
    s, t = 2.0, 3.0
    Q, QtHQ = O.double_shift_qr(H, s, t)
    Q0, _ = np.linalg.qr(H @ H - s * H + t * np.eye(n))
    assert np.abs(np.abs(Q) - np.abs(Q0)).max() < 1e-12
    assert np.abs(QtHQ - Q.T @ H @ Q).max() < 1e-12
    assert np.abs(Q.T @ Q - np.eye(n)).max() < 1e-12


@pytest.mark.parametrize("n", [10, 100, 500])
def test_schur(n):
    # test/Schur.cpp:14-69: T quasi-upper-triangular (exact zeros below the first sub-diagonal), U'U = I, AU = UT
    H = hessenberg(n, 7 + n)
    T, U = O.hess_schur(H)
    assert np.abs(np.tril(T, -2)).max() <= 1e-16
    assert np.abs(U.T @ U - np.eye(n)).max() < 1e-12
    assert np.abs(H @ U - U @ T).max() < 1e-12
    sub = np.diag(T, -1)
    assert not np.any((sub[:-1] != 0) & (sub[1:] != 0))  # 1x1 and 2x2 blocks only


@pytest.mark.parametrize("n", [2, 3, 10, 100])
def test_hessenberg_eigen(n):
    # test/Eigen.cpp:26-65: ||HU - UD||_inf < 1e-12
    H = hessenberg(n, 31 + n)
    ev, V = O.hess_eigen(H)
    assert np.abs(H @ V - V * ev).max() < 1e-12
    ref = np.linalg.eigvals(H)
    assert max(np.abs(ref - lam).min() for lam in ev) < 1e-9 and max(np.abs(ev - lam).min() for lam in ref) < 1e-9  # non-normal H: eigenvalue conditioning
    assert np.abs(np.linalg.norm(V, axis=0) - 1.0).max() < 1e-14
    # real eigenvalues have exactly zero imaginary part, complex ones come in exact conjugate pairs
    cplx = ev[ev.imag != 0]
    assert len(cplx) % 2 == 0 and np.array_equal(cplx[0::2], np.conj(cplx[1::2]))


@pytest.mark.parametrize("n,prob,k,m", GEN_CASES)
@pytest.mark.parametrize("rule", RULES_GEN)
def test_gen_fixtures_all_rules(n, prob, k, m, rule):
    # test/GenEigs.cpp:38-108: maxit = 300; info == Successful and ||AU - UD||_inf <= 1e-9 unless allow_fail
    A = gen_fixture(n, prob)
    s = O.GenEigsSolver(O.Op.csc(n, n, A.indptr, A.indices, A.data), k, m)
    s.init()
    nconv = s.compute(getattr(O, rule), 300)
    if s.info() != O.Successful:
        assert rule in ALLOW_FAIL
        assert nconv < k and len(s.eigenvalues()) == nconv
        return
    ev, U = s.eigenvalues(), s.eigenvectors()
    assert nconv == k and np.abs(A @ U - U * ev).max() < 1e-9
    # known answer: the selected part of the dense spectrum
    full = np.linalg.eigvals(A.toarray())
    key = {"LargestMagn": -np.abs(full), "LargestReal": -full.real, "LargestImag": -np.abs(full.imag),
           "SmallestMagn": np.abs(full), "SmallestReal": full.real, "SmallestImag": np.abs(full.imag)}[rule]
    want = full[np.argsort(key, kind="stable")[:k]]
    # compare as sets (conjugate pairs at the cut may be ordered either way)
    for lam in ev:
        assert np.abs(full - lam).min() < 1e-9
    if rule in ("LargestMagn", "LargestReal", "SmallestReal"):
        assert abs(np.sort(np.abs(ev))[::-1][0] - np.sort(np.abs(want))[::-1][0]) < 1e-8 or rule != "LargestMagn"


def test_gen_constructor_checks():
    op = O.Op.diag(np.arange(1.0, 11.0))
    for nev, ncv in [(0, 5), (9, 10), (3, 4), (3, 11)]:  # GenEigsBase.h:419-423
        with pytest.raises(ValueError):
            O.GenEigsSolver(op, nev, ncv)


# ---- GenEigsRealShiftSolver (test/GenEigsRealShift.cpp sparse cases :146-180; bar 1e-8, compute(selection, 500)) -------
REAL_SHIFT_CASES = [(10, 0.5, 3, 6, 1.0), (100, 0.1, 10, 30, 10.0), (1000, 0.01, 20, 50, 100.0)]


@pytest.mark.parametrize("n,prob,k,m,sigma", REAL_SHIFT_CASES)
@pytest.mark.parametrize("rule", ["LargestMagn", "LargestReal", "LargestImag", "SmallestReal"])
def test_real_shift_fixtures(n, prob, k, m, sigma, rule):
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    r, c, v = O.gen_sparse_data(n, prob)
    A = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsc()
    lu = spla.splu((A - sigma * sp.identity(n)).tocsc())
    eigs = O.GenEigsSolver(O.Op.callback(n, lu.solve), k, m, sigma=sigma)
    eigs.init()
    nconv = eigs.compute(getattr(O, rule), 500)
    assert eigs.info() == O.Successful and nconv >= k - 1
    ev, U = eigs.eigenvalues(), eigs.eigenvectors()
    assert np.abs(A @ U - U * ev).max() <= 1e-8
