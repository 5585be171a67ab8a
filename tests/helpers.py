"""Shared fixtures of the parity tests: the reference's own matrices, rebuilt without Eigen."""
import numpy as np
import scipy.sparse as sp

import oracle

RULES_SYM = ["LargestMagn", "LargestAlge", "SmallestMagn", "SmallestAlge", "BothEnds"]
SPARSE_CASES = [(10, 0.5, 3, 6), (100, 0.1, 10, 20), (1000, 0.01, 20, 50)]  # test/SymEigs.cpp:133-167


def sparse_fixture(n, prob):
    """gen_sparse_data(n, prob) of test/SymEigs.cpp:25-42 as a scipy CSC matrix (NOT symmetric: the
    solver is only allowed to read its lower triangle) plus the symmetric matrix that triangle defines."""
    r, c, v = oracle.gen_sparse_data(n, prob)
    A = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsc()
    A.sort_indices()
    S = (sp.tril(A) + sp.tril(A, -1).T).tocsr()
    return A, S


def cycle_laplacian(n):
    """test/Example1.cpp:18-32"""
    L = np.zeros((n, n))
    for i in range(n):
        L[i, i] = 1.0
        L[i, (i + n - 1) % n] = -0.5
        L[i, (i + 1) % n] = -0.5
    return L


EXAMPLE2 = [  # test/Example2.cpp:54-58, :66-70, :77-81 (issue #159: near rank-1 5x5 matrices)
    np.array([[15.035447086947079479, 3.932587856183598677, -4.848070276813470542, -8.027254936523050904, -2.865327349780228231],
              [3.932587856183598677, 1.028585791773944732, -1.268034278346991263, -2.099564123322002035, -0.749439073848281425],
              [-4.848070276813470542, -1.268034278346991263, 1.563224909309606855, 2.588329820664053864, 0.923903910371237535],
              [-8.027254936523050904, -2.099564123322002035, 2.588329820664053864, 4.285660509016328222, 1.529765824738644411],
              [-2.865327349780228231, -0.749439073848281425, 0.923903910371237535, 1.529765824738644411, 0.546049663433429209]]),
    np.array([[0.6118330552, -3.058379358, 1.329013596, 2.601267208, 1.072783220],
              [-3.058379358, 15.28796821, -6.643360824, -13.00299463, -5.362538075],
              [1.329013596, -6.643360824, 2.886861251, 5.650429406, 2.330281884],
              [2.601267208, -13.00299463, 5.650429406, 11.05953826, 4.561041261],
              [1.072783220, -5.362538075, 2.330281884, 4.561041261, 1.881009576]]),
    np.array([[17.7699571312182, 10.7033479738827, -19.1658731825582, -4.20053658859459, -11.1426294187651],
              [10.7033479738827, 6.44692933157151, -11.5441477084849, -2.53010203979439, -6.71152097511499],
              [-19.1658731825582, -11.5441477084849, 20.6714451890590, 4.53050904744533, 12.0179368348118],
              [-4.20053658859459, -2.53010203979439, 4.53050904744533, 0.992940360059961, 2.63394122006329],
              [-11.1426294187651, -6.71152097511499, 12.0179368348118, 2.63394122006329, 6.98697185632535]]),
]


def random_tridiag(n, seed):
    rng = np.random.default_rng(seed)
    d = rng.uniform(-1, 1, n)
    e = rng.uniform(-1, 1, n - 1)
    return np.diag(d) + np.diag(e, -1) + np.diag(e, 1)


def wanted_by_rule(evals_all, rule, k):
    """The k eigenvalues rule `rule` selects from the full spectrum (ascending input)."""
    ev = np.sort(evals_all)
    if rule == "LargestAlge":
        return np.sort(ev[-k:])
    if rule == "SmallestAlge":
        return np.sort(ev[:k])
    if rule == "LargestMagn":
        return np.sort(ev[np.argsort(-np.abs(ev), kind="stable")[:k]])
    if rule == "SmallestMagn":
        return np.sort(ev[np.argsort(np.abs(ev), kind="stable")[:k]])
    if rule == "BothEnds":  # more from the high end when k is odd (SelectionRule.h:56-57)
        hi = (k + 1) // 2
        return np.sort(np.concatenate([ev[-hi:], ev[:k - hi]]))
    raise ValueError(rule)


def check_window_records(T, S, row_begin=0):
    """Invariants of the x-window records of the int32 CSR kernel (include/mispec.h mispec_csr_windows_table) against the stored
    matrix S (rows of a shard that starts at global row row_begin; global column indices): windows sorted, disjoint, 128-byte
    aligned, LDS positions consecutive and even; every entry of a block without the far flag inside a window; the covered count
    exact.  Returns the number of covered entries."""
    import numpy as np

    S = S.tocsr()
    S.sort_indices()
    n = S.shape[0]
    covered = 0
    assert T.shape == ((n + 255) // 256, 32)
    for b in range(T.shape[0]):
        rec = T[b]
        nw, far, total = int(rec[0] & 255), int(rec[0] >> 8), int(rec[1])
        st, ad, en = rec[4:12].astype(np.int64), rec[12:20].astype(np.int64), rec[20:28].astype(np.int64)
        cols = S.indices[S.indptr[b * 256]: S.indptr[min(n, (b + 1) * 256)]].astype(np.int64)
        inside = np.zeros(cols.size, bool)
        base = 0
        assert 0 <= nw <= 8 and total <= 6144, (b, rec)
        for w in range(nw):
            assert st[w] % 16 == 0 and st[w] + ad[w] == base and en[w] > st[w] and (w == 0 or st[w] >= en[w - 1]), (b, rec)
            base += en[w] - st[w]
            inside |= (cols >= st[w]) & (cols < en[w])
        assert base == total and total % 2 == 0 and np.all(st[nw:] == 0x3FFFFFFF), (b, rec)
        assert int(inside.sum()) == rec[2] and (far or inside.all()), (b, rec)
        assert nw > 0 or far, (b, rec)
        covered += int(inside.sum())
    return covered
