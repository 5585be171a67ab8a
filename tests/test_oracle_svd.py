"""contrib/PartialSVDSolver.h on the oracle, pinned the way test/SVD.cpp:35-67 pins the reference: against a full
dense SVD, |sigma - sigma_ref| <= 1e-9 and | |U| - |U_ref| |, | |V| - |V_ref| | <= 1e-9, nconv == k — on the
reference's own reproducible sparse fixtures (test/SVD.cpp:17-33, tall 1000x100 and wide 100x1000, prob 0.1,
k = 5, ncv = 10; the dense MatrixXd::Random cases cannot be regenerated without Eigen)."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle as O


def svd_fixture(m, n, prob=0.1):
    r, c, v = O.gen_sparse_data_rect(m, n, prob)
    return sp.coo_matrix((v, (r, c)), shape=(m, n)).tocsr()


@pytest.mark.parametrize("shape", [(1000, 100), (100, 1000)])
def test_partial_svd_matches_dense_svd(shape):
    A = svd_fixture(*shape)
    k, ncv = 5, 10
    nconv, sv, U, V = O.partial_svd(A, k, ncv)
    assert nconv == k
    Ur, sr, Vtr = np.linalg.svd(A.toarray(), full_matrices=False)
    assert np.abs(sv - sr[:k]).max() <= 1e-9
    assert np.abs(np.abs(U) - np.abs(Ur[:, :k])).max() <= 1e-9
    assert np.abs(np.abs(V) - np.abs(Vtr[:k].T)).max() <= 1e-9


def test_fixture_is_the_reference_generator():
    # same engine and draw order as gen_sparse_data(n, prob) of test/SymEigs.cpp for a square shape
    r0, c0, v0 = O.gen_sparse_data(50, 0.2)
    r1, c1, v1 = O.gen_sparse_data_rect(50, 50, 0.2)
    assert np.array_equal(r0, r1) and np.array_equal(c0, c1) and np.array_equal(v0, v1)
