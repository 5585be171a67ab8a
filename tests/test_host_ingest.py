"""Host stages of the matrix ingest (no device needed): the triangle -> full symmetric CSR mirroring that SparseSymMatProd /
SparseSymShiftSolve / SparseRegularInverse start with (MatOp/SparseSymMatProd.h:83-88 reads one triangle through
selfadjointView<Uplo>), run by the library's host threads (common.hpp ingest_threads).  Integer work: checked exactly."""
import numpy as np
import pytest
import scipy.sparse as sp

import spectra_amd as sa
from helpers import sparse_fixture


def reference_mirror(M, uplo):
    T = sp.tril(M) if uplo == "L" else sp.triu(M)
    D = sp.diags(T.diagonal())
    S = (T + T.T - D).tocsr()
    S.sort_indices()
    return S


@pytest.mark.parametrize("uplo", ["L", "U"])
@pytest.mark.parametrize("fmt", ["csc", "csr"])
def test_mirror_equals_scipy_on_the_reference_fixture(uplo, fmt):
    A, _ = sparse_fixture(1000, 0.01)          # NOT symmetric: only the requested triangle may be read
    M = A.asformat(fmt)
    got = sa.mirror_triangle_host(M, uplo)
    ref = reference_mirror(A, uplo)
    assert np.array_equal(got.indptr, ref.indptr) and np.array_equal(got.indices, ref.indices)
    assert np.array_equal(got.data, ref.data)


def test_mirror_with_many_threads_scattered_columns_and_unsorted_input():
    # enough entries for every host thread to take part; explicit zeros and a zero diagonal are kept; unsorted input rows are
    # sorted on the way out
    n = 400_000
    rng = np.random.default_rng(3)
    rows = np.repeat(np.arange(n), 4)
    cols = rng.integers(0, n, rows.size)
    vals = rng.uniform(-1, 1, rows.size)
    vals[::97] = 0.0
    M = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsc()
    M.sum_duplicates()
    got = sa.mirror_triangle_host(M, "L")
    ref = reference_mirror(M, "L")
    # scipy drops nothing either: compare as matrices, then structure
    assert abs(got - ref).max() == 0.0
    for i in (0, 1, n // 2, n - 1):
        seg = got.indices[got.indptr[i]:got.indptr[i + 1]]
        assert np.all(np.diff(seg) > 0)
    # unsorted storage order inside the columns
    Mu = M.copy()
    for j in range(0, 2000):
        s, e = Mu.indptr[j], Mu.indptr[j + 1]
        p = rng.permutation(e - s)
        Mu.indices[s:e] = Mu.indices[s:e][p]
        Mu.data[s:e] = Mu.data[s:e][p]
    Mu.has_sorted_indices = False
    got_u = sa.mirror_triangle_host(Mu, "L")
    assert np.array_equal(got_u.indptr, got.indptr) and np.array_equal(got_u.indices, got.indices) and np.array_equal(got_u.data, got.data)


def test_mirror_reports_bad_indices():
    M = sp.csc_matrix((np.ones(2), np.array([0, 1]), np.array([0, 1, 2])), shape=(2, 2))
    M.indices = np.array([0, 7], dtype=np.int32)
    with pytest.raises(ValueError, match="out of range"):
        sa.mirror_triangle_host(M, "L")
