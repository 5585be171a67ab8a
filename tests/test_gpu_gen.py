"""a7 / a18 / a19 — GenEigsSolver (implicitly-restarted Arnoldi) on the GPU through the C ABI vs the CPU oracle,
on the reference's own solver tests (test/GenEigs.cpp:38-174) and on the config-4 style benchmark matrix."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle as O
import spectra_amd as sa

pytestmark = pytest.mark.gpu

GEN_CASES = [(10, 0.5, 3, 6), (100, 0.1, 10, 30), (1000, 0.01, 20, 50)]
RULES_GEN = ["LargestMagn", "LargestReal", "LargestImag", "SmallestMagn", "SmallestReal", "SmallestImag"]
ALLOW_FAIL = {"SmallestMagn", "SmallestImag"}  # test/GenEigs.cpp:98,106


def gen_fixture(n, prob):
    r, c, v = O.gen_sparse_data(n, prob)
    A = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsc()
    A.sort_indices()
    return A


def match_sets(a, b, tol):
    """every value of a has a partner in b within tol (conjugate pairs may come in either order)"""
    return all(np.abs(b - x).min() < tol for x in a) and all(np.abs(a - x).min() < tol for x in b)


@pytest.mark.parametrize("n,prob,k,m", GEN_CASES)
@pytest.mark.parametrize("rule", RULES_GEN)
def test_gen_fixtures_all_rules(ctx, n, prob, k, m, rule):
    A = gen_fixture(n, prob)
    op = sa.SparseGenMatProd(A, ctx=ctx)          # ColMajor input, the reference default
    eigs = sa.GenEigsSolver(op, k, m)
    eigs.init()
    nconv = eigs.compute(sa.SortRule[rule], 300)   # test/GenEigs.cpp:44 maxit = 300
    ref = O.GenEigsSolver(O.Op.csc(n, n, A.indptr, A.indices, A.data), k, m)
    ref.init()
    o_nconv = ref.compute(getattr(O, rule), 300)
    if eigs.info() != sa.CompInfo.Successful:
        assert rule in ALLOW_FAIL and ref.info() != O.Successful   # fails exactly where the reference may fail
        assert len(eigs.eigenvalues()) == nconv < k
        return
    assert ref.info() == O.Successful and nconv == o_nconv == k
    evals, evecs = eigs.eigenvalues(), eigs.eigenvectors()
    assert np.abs(A @ evecs - evecs * evals).max() < 1e-9          # test/GenEigs.cpp:67-70
    assert match_sets(evals, ref.eigenvalues(), 1e-9)
    assert abs(eigs.num_operations() - ref.num_operations()) <= max(3 * m, 0.15 * ref.num_operations())
    res = eigs.residuals()                                          # evaluated on the device
    host = np.linalg.norm(A @ evecs - evecs * evals, axis=0) / np.linalg.norm(evecs, axis=0)
    assert np.abs(res - host).max() < 1e-12


def test_gen_user_operator_and_errors(ctx):
    rng = np.random.default_rng(5)
    M = rng.uniform(-1, 1, (60, 60))

    class Op:
        def rows(self):
            return 60

        def cols(self):
            return 60

        def perform_op(self, x):
            return M @ x

    eigs = sa.GenEigsSolver(Op(), 6, 20, ctx=ctx)
    eigs.init()
    assert eigs.compute(sa.SortRule.LargestMagn) == 6 and eigs.info() == sa.CompInfo.Successful
    ev, U = eigs.eigenvalues(), eigs.eigenvectors()
    assert np.abs(M @ U - U * ev).max() < 1e-9
    full = np.linalg.eigvals(M)
    assert all(np.abs(full - x).min() < 1e-9 for x in ev)
    assert np.all(np.diff(np.abs(ev)) <= 1e-12)           # default sorting LargestMagn
    for nev, ncv in [(0, 5), (59, 60), (3, 4), (3, 61)]:   # GenEigsBase.h:419-423
        with pytest.raises(ValueError, match="must satisfy"):
            sa.GenEigsSolver(Op(), nev, ncv, ctx=ctx)
    with pytest.raises(ValueError):                        # symmetric-only rule (SelectionRule.h:72-80)
        eigs.init()
        eigs.compute(sa.SortRule.LargestAlge)


@pytest.mark.parametrize("n", [200_000, 5_000_000])
def test_config4_nonsymmetric_band(ctx, n):
    # BASELINE.json configs[3]: GenEigsSolver on a 5M x 5M non-symmetric CSR (~15 nnz/row), k = 10, ncv = 30.
    if n == 5_000_000:  # full size: against the oracle's complete solve (tests/golden/full_size_c4.json)
        from test_gpu_fullsize import check_c4_solve

        return check_c4_solve(ctx)
    op = sa.SparseGenMatProd.synth_band(n, ctx=ctx)
    eigs = sa.GenEigsSolver(op, 10, 30)
    eigs.init()
    nconv = eigs.compute(sa.SortRule.LargestMagn, 1000, 1e-11)
    assert nconv == 10 and eigs.info() == sa.CompInfo.Successful
    res = eigs.residuals()
    assert res.max() <= 1e-10, res
    evals = eigs.eigenvalues()
    assert np.all(np.diff(np.abs(evals)) <= 1e-12)
    if n <= 200_000:
        rp, ci, v = O.synth_band_csr(n, symmetric=False)
        ref = O.GenEigsSolver(O.Op.csr(n, n, rp, ci, v), 10, 30)
        ref.init()
        assert ref.compute(O.LargestMagn, 1000, 1e-11) == 10
        assert match_sets(evals, ref.eigenvalues(), 1e-9)
        A = sp.csr_matrix((v, ci, rp), shape=(n, n))
        X = eigs.eigenvectors()
        assert (np.linalg.norm(A @ X - X * evals, axis=0) / np.linalg.norm(X, axis=0)).max() <= 1e-10


def test_device_driven_arnoldi_steps_equal_host_driven_steps():
    # Device-driven Arnoldi steps (h, |h|, beta and the 0.717 test kept in device memory) must take exactly the
    # decisions of the host-synchronous path: same kernels in the same order => bit-identical results and counters,
    # with far fewer host synchronisations.
    import os
    import subprocess
    import sys

    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); import spectra_amd as sa\n"
        "op = sa.SparseGenMatProd.synth_band(200000)\n"
        "e = sa.GenEigsSolver(op, 8, 24); e.init(); n = e.compute(sa.SortRule.LargestMagn, 1000, 1e-11)\n"
        "print(n, e.num_operations(), e.num_iterations(), e.eigenvalues().tobytes().hex(), e.get_profile()['n_host_sync'])\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for host in ("0", "1"):
        env = dict(os.environ, MISPEC_HOST_STEPS=host)
        r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout.split())
    assert outs[0][:4] == outs[1][:4]
    assert float(outs[0][4]) < 0.5 * float(outs[1][4])


def test_gen_wide_basis(ctx):
    # GenEigsSolver with ncv = 80 (> 64 columns: panelled Arnoldi orthogonalisation)
    from helpers import sparse_fixture

    n, nev, ncv = 1000, 30, 80
    A, _ = sparse_fixture(n, 0.01)
    eigs = sa.GenEigsSolver(sa.SparseGenMatProd(A, ctx=ctx), nev, ncv)
    eigs.init()
    nconv = eigs.compute(sa.SortRule.LargestMagn, 1000, 1e-10)
    assert nconv >= nev - 1  # a conjugate pair may be split at the nev boundary
    ev, U = eigs.eigenvalues(), eigs.eigenvectors()
    assert np.abs(A @ U - U * ev).max() <= 1e-9


def test_gen_very_wide_basis(ctx):
    # ncv = 150 > 128 columns
    from helpers import sparse_fixture

    n, nev, ncv = 1000, 60, 150
    A, _ = sparse_fixture(n, 0.01)
    eigs = sa.GenEigsSolver(sa.SparseGenMatProd(A, ctx=ctx), nev, ncv)
    eigs.init()
    nconv = eigs.compute(sa.SortRule.LargestMagn, 1000, 1e-10)
    assert nconv >= nev - 1
    ev, U = eigs.eigenvalues(), eigs.eigenvectors()
    assert np.abs(A @ U - U * ev).max() <= 1e-9


# ---- GenEigsRealShiftSolver + SparseGenRealShiftSolve (GenEigsRealShiftSolver.h; test/GenEigsRealShift.cpp:146-180) -----
REAL_SHIFT_CASES = [(10, 0.5, 3, 6, 1.0), (100, 0.1, 10, 30, 10.0), (1000, 0.01, 20, 50, 100.0)]


@pytest.mark.parametrize("n,prob,k,m,sigma", REAL_SHIFT_CASES)
@pytest.mark.parametrize("rule", ["LargestMagn", "LargestReal", "LargestImag", "SmallestReal"])
def test_real_shift_fixtures(ctx, n, prob, k, m, sigma, rule):
    import scipy.sparse.linalg as spla
    from helpers import sparse_fixture

    A, _ = sparse_fixture(n, prob)
    op = sa.SparseGenRealShiftSolve(A, ctx=ctx)
    eigs = sa.GenEigsRealShiftSolver(op, k, m, sigma)
    eigs.init()
    nconv = eigs.compute(sa.SortRule[rule], 500)
    assert eigs.info() == sa.CompInfo.Successful and nconv >= k - 1  # a conjugate pair may be split at the nev boundary
    ev, U = eigs.eigenvalues(), eigs.eigenvectors()
    assert np.abs(A @ U - U * ev).max() <= 1e-8            # test/GenEigsRealShift.cpp:66-70
    lu = spla.splu((A - sigma * sp.identity(n)).tocsc())
    oe = O.GenEigsSolver(O.Op.callback(n, lu.solve), k, m, sigma=sigma)
    oe.init()
    oe.compute(getattr(O, rule), 500)
    ev0 = oe.eigenvalues()
    assert len(ev0) == len(ev)
    assert all(np.abs(ev0 - lam).min() <= 1e-7 * max(1.0, abs(lam)) for lam in ev)


def test_real_shift_operator(ctx):
    from helpers import sparse_fixture

    A, _ = sparse_fixture(100, 0.1)
    op = sa.SparseGenRealShiftSolve(A, ctx=ctx)
    op.set_shift(10.0)
    x = np.random.default_rng(5).uniform(-1, 1, 100)
    ref = np.linalg.solve(A.toarray() - 10.0 * np.eye(100), x)
    assert np.abs(op.perform_op(x) - ref).max() <= 1e-12
    with pytest.raises(ValueError, match="4096"):
        sa.SparseGenRealShiftSolve(sp.identity(5000, format="csc"), ctx=ctx)


# ---- GenEigsComplexShiftSolver (GenEigsComplexShiftSolver.h:20-150) ----------------------------------------------------
@pytest.mark.parametrize("n,prob,k,m,sr,si", [(10, 0.5, 3, 6, 2.0, 1.0), (100, 0.1, 10, 30, 20.0, 10.0), (1000, 0.01, 20, 50, 200.0, 100.0)])
@pytest.mark.parametrize("rule", ["LargestMagn", "LargestReal", "LargestImag", "SmallestReal"])
def test_complex_shift_reference_fixtures(ctx, n, prob, k, m, sr, si, rule):
    # test/GenEigsComplexShift.cpp:150-186 x :75-108; bar ||AU - UD||_inf < 1e-8 (:71)
    import scipy.sparse as sp

    r, c, v = O.gen_sparse_data(n, prob)
    A = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsc()
    op = sa.SparseGenComplexShiftSolve(A, ctx=ctx)
    eigs = sa.GenEigsComplexShiftSolver(op, k, m, sr, si)
    eigs.init()
    nconv = eigs.compute(sa.SortRule[rule])
    assert eigs.info() == sa.CompInfo.Successful and nconv > 0
    evals, evecs = eigs.eigenvalues(), eigs.eigenvectors()
    Ad = A.toarray()
    assert np.abs(Ad @ evecs - evecs * evals).max() < 1e-8
    if n <= 100:  # oracle parity (numpy complex inverse as the oracle's operator)
        Minv = np.linalg.inv(Ad - (sr + 1j * si) * np.eye(n))
        Pinv = np.linalg.inv(Ad - O.complex_shift_probe(sr) * np.eye(n))
        ref = O.GenEigsSolver(O.Op.callback(n, lambda x: (Minv @ x).real), k, m,
                              complex_shift=(sr, si, O.Op.callback(n, lambda x: Pinv @ x)))
        ref.init()
        ref.compute(getattr(O, rule))
        assert len(ref.eigenvalues()) == len(evals)
        # a conjugate pair cut by the nev boundary may be represented by either member: compare up to conjugation
        fold = lambda z: np.sort_complex(z.real + 1j * np.abs(z.imag))
        assert np.abs(fold(ref.eigenvalues()) - fold(evals)).max() < 1e-8


def test_complex_shift_operator_and_dense_form(ctx):
    # MatOp/SparseGenComplexShiftSolve.h:100-111 / DenseGenComplexShiftSolve.h:85-102: y = Re((A - sigma I)^{-1} x)
    n = 60
    A = np.random.default_rng(4).uniform(-1, 1, (n, n))
    x = np.random.default_rng(5).uniform(-1, 1, n)
    for op in (sa.DenseGenComplexShiftSolve(A, ctx=ctx), sa.SparseGenComplexShiftSolve(sp.csc_matrix(A), ctx=ctx)):
        op.set_shift(0.3, 0.7)
        y = op.perform_op(x)
        ref = np.linalg.solve(A - (0.3 + 0.7j) * np.eye(n), x).real
        assert np.abs(y - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())
        op.set_shift(0.3, 0.0)  # a real shift through the same entry point
        assert np.abs(op.perform_op(x) - np.linalg.solve(A - 0.3 * np.eye(n), x)).max() < 1e-10
    eigs = sa.GenEigsComplexShiftSolver(sa.DenseGenComplexShiftSolve(A, ctx=ctx), 6, 30, 0.3, 0.7)
    eigs.init()
    assert eigs.compute(sa.SortRule.LargestMagn) >= 5
    ev, U = eigs.eigenvalues(), eigs.eigenvectors()
    assert np.abs(A @ U - U * ev).max() < 1e-8
