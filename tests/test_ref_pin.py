"""What pins the oracle: the REFERENCE'S OWN CODE.

``oracle/_ref/libspectra_ref.so`` is yixuan/spectra's headers (SymEigsSolver, HermEigsBase, Lanczos, Arnoldi, TridiagQR,
TridiagEigen, Givens, SparseSymMatProd, GenEigsSolver, DoubleShiftQR, UpperHessenbergEigen ...) compiled unmodified, where they
lie, by ``oracle/build_ref.sh`` — ``oracle/eigen_shim`` (a small dense/sparse algebra with Eigen's names, NOT Eigen) stands in for
the one dependency this image lacks.  The shim fixes the one thing Eigen leaves unspecified — the order of its reductions — to
the order the restatement uses (left to right), so the two can be compared BIT FOR BIT: every branch, every counter, every
scalar expression of ``oracle/spectra_oracle*.hpp`` is checked against the reference's control flow
(HermEigsBase.h:366-390, Lanczos.h:62-187, Arnoldi.h:198-340, UpperHessenbergQR.h:515-693, TridiagEigen.h:121-210,
GenEigsBase.h:43-340, DoubleShiftQR.h:51-438, UpperHessenbergEigen.h:53-320, UpperHessenbergSchur.h:46-425).

Two layers:
  * tests that need the library (skipped where neither ``/root/reference`` nor a prebuilt ``oracle/_ref`` is present);
  * ``test_restatement_equals_the_committed_reference_vectors`` — always runs: the oracle against
    ``tests/golden/ref_pin_golden.npz``, which ``tests/golden/make_ref_golden.py`` wrote from ``oracle/_ref``.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import oracle as O
from helpers import EXAMPLE2, RULES_SYM, SPARSE_CASES, cycle_laplacian, random_tridiag, sparse_fixture
from oracle import ref as R

RULES_GEN = ["LargestMagn", "LargestReal", "LargestImag", "SmallestMagn", "SmallestReal", "SmallestImag"]
SHIFT_SYM = [(10, 0.5, 3, 6, 1.0), (100, 0.1, 10, 20, 10.0), (1000, 0.01, 20, 50, 100.0)]   # test/SymEigsShift.cpp:148-185
SHIFT_GEN = [(10, 0.5, 3, 6, 1.0), (100, 0.1, 10, 30, 10.0), (1000, 0.01, 20, 50, 100.0)]   # test/GenEigsRealShift.cpp:146-180
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_pin_golden.npz"))
needs_ref = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built and /root/reference absent")


def oracle_sym(op, k, m, rule, tol=1e-10, sorting=O.LargestAlge, v0=None, maxit=1000):
    s = O.SymEigsSolver(op, k, m)
    s.init(v0)
    nconv = s.compute(getattr(O, rule), maxit, tol, sorting)
    return [nconv, s.info(), s.num_iterations(), s.num_operations()], s.eigenvalues(), s.eigenvectors()


def oracle_gen(op, k, m, rule, v0=None):
    s = O.GenEigsSolver(op, k, m)
    s.init(v0)
    nconv = s.compute(getattr(O, rule))
    return [nconv, s.info(), s.num_iterations(), s.num_operations()], s.eigenvalues(), s.eigenvectors()


def counters(r):
    return [r.nconv, r.info, r.num_iterations, r.num_operations]


# ---- always: the restatement against what the reference's code returned ----------------------------------------------------------
def test_restatement_equals_the_committed_reference_vectors():
    assert b"yixuan/spectra" in GOLD["describe"].tobytes()
    for n, prob, k, m in SPARSE_CASES:
        A, S = sparse_fixture(n, prob)
        for rule in RULES_SYM:
            c, ev, _ = oracle_sym(O.Op.csc_sym(n, A.indptr, A.indices, A.data, True), k, m, rule)
            assert c == list(GOLD[f"sym_{n}_{rule}"]), (n, rule)
            assert np.abs(ev - GOLD[f"sym_{n}_{rule}_evals"]).max() <= 1e-12
            assert np.array_equal(ev, GOLD[f"sym_{n}_{rule}_evals"])  # in fact: the same bits
        for rule in RULES_GEN:
            c, ev, _ = oracle_gen(O.Op.csc(n, n, A.indptr, A.indices, A.data), k, m, rule)
            assert c == list(GOLD[f"gen_{n}_{rule}"]), (n, rule)
            assert np.array_equal(ev, GOLD[f"gen_{n}_{rule}_evals"].view(np.complex128))
    M = cycle_laplacian(20)
    for k, m in [(3, 6), (5, 12), (6, 12)]:
        c, ev, _ = oracle_sym(O.Op.dense_sym(M), k, m, "LargestMagn", tol=1e-15, sorting=O.SmallestAlge)
        assert c == list(GOLD[f"example1_{k}_{m}"]) and np.array_equal(ev, GOLD[f"example1_{k}_{m}_evals"])
    for i, M2 in enumerate(EXAMPLE2):
        c, ev, _ = oracle_sym(O.Op.dense_sym(M2), 1, 3, "LargestAlge")
        assert c == list(GOLD[f"example2_{i}"]) and np.array_equal(ev, GOLD[f"example2_{i}_evals"])
    c, ev, _ = oracle_sym(O.Op.diag(np.arange(1.0, 11.0)), 3, 6, "LargestAlge")
    assert c == list(GOLD["doc_diag"]) and np.abs(ev - GOLD["doc_diag_evals"]).max() <= 1e-12
    # the shift-and-invert drivers: same counters; the eigenvalues to 1e-10 only — the shift solve behind both sides is scipy's
    # sparse LU, whose bits belong to the scipy build, not to either code (the bit-for-bit comparison is the @needs_ref test)
    import scipy.sparse.linalg as spla

    for n, prob, k, m, sigma in SHIFT_SYM:
        A, S = sparse_fixture(n, prob)
        lu = spla.splu((S - sigma * sp.identity(n)).tocsc())
        for rule in RULES_SYM:
            o = O.SymEigsSolver(O.Op.callback(n, lu.solve), k, m, sigma=sigma)
            o.init()
            nconv = o.compute(getattr(O, rule))
            g = list(GOLD[f"symshift_{n}_{rule}"])
            assert [nconv, o.info()] == g[:2] and abs(o.num_operations() - g[3]) <= (m - k), (n, rule)
            if g[1] == 0:
                assert np.abs(o.eigenvalues() - GOLD[f"symshift_{n}_{rule}_evals"]).max() <= 1e-10 * max(1.0, sigma)


# ---- with the library: unit by unit, then whole solves ---------------------------------------------------------------------------
@needs_ref
def test_reference_library_is_the_reference():
    d = R.lib().ref_describe().decode()
    assert "yixuan/spectra" in d and "1.2.0" in d and "not Eigen" in d


@needs_ref
def test_simple_random_and_argsort():
    for seed in (0, 1, 123, 2 ** 31 - 1, 2 ** 31 + 5):
        assert np.array_equal(O.simple_random(1000, seed), R.simple_random(1000, seed))  # Util/SimpleRandom.h:30-123
    rng = np.random.default_rng(5)
    v = rng.uniform(-1, 1, 37)
    v[5] = v[9]  # a tie
    for rule in (O.LargestMagn, O.LargestAlge, O.SmallestMagn, O.SmallestAlge, O.BothEnds):
        assert np.array_equal(O.argsort(rule, v), R.argsort(rule, v))  # Util/SelectionRule.h:195-287


@needs_ref
def test_givens_rotation_bit_for_bit():
    # LinAlg/Givens.h:149-206 incl. the small-ratio Taylor branch and the zero cases
    vals = [0.0, 1.0, -1.0, 1e-300, -3e-9, 1e-5, 2.5e-4, 0.3, -0.7, 12.0, 1e150, -1e-160]
    for x in vals:
        for y in vals:
            a, b = O.givens(x, y), R.givens(x, y)
            assert np.array_equal(np.array(a), np.array(b), equal_nan=True), (x, y, a, b)


@needs_ref
@pytest.mark.parametrize("n", [2, 3, 6, 21, 40])
def test_tridiag_qr_and_eigen_bit_for_bit(n):
    for seed in range(3):
        T = random_tridiag(n, 100 * n + seed)
        if seed == 2 and n > 3:
            T[2, 1] = T[1, 2] = 1e-18  # deflation branch, UpperHessenbergQR.h:533-539
        for shift in (0.0, 0.37, float(T[n - 1, n - 1])):
            for a, b in zip(O.tridiag_qr(T, shift), R.tridiag_qr(T, shift)):  # :515-598, :627-693, :383-417
                assert np.array_equal(a, b)
        for a, b in zip(O.tridiag_eigen(T), R.tridiag_eigen(T)):  # LinAlg/TridiagEigen.h:44-210
            assert np.array_equal(a, b)


@needs_ref
@pytest.mark.parametrize("n", [3, 4, 9, 30])
def test_hessenberg_qr_double_shift_and_eigen_bit_for_bit(n):
    rng = np.random.default_rng(n)
    for trial in range(3):
        H = np.triu(rng.uniform(-1, 1, (n, n)), -1)
        if trial == 2 and n > 4:
            H[3, 2] = 0.0  # a split, DoubleShiftQR.h:155-170
        for a, b in zip(O.hess_qr(H, 0.21), R.hess_qr(H, 0.21)):  # UpperHessenbergQR.h:136-255
            assert np.array_equal(a, b)
        for a, b in zip(O.double_shift_qr(H, 0.4, 0.3), R.double_shift_qr(H, 0.4, 0.3)):  # DoubleShiftQR.h:51-438
            assert np.array_equal(a, b)
        for a, b in zip(O.hess_eigen(H), R.hess_eigen(H)):  # UpperHessenbergEigen.h:53-320 + UpperHessenbergSchur.h
            assert np.array_equal(a, b)


@needs_ref
@pytest.mark.parametrize("n,prob,k,m", SPARSE_CASES)
def test_operators_and_factorisations_bit_for_bit(n, prob, k, m):
    A, S = sparse_fixture(n, prob)
    x = O.simple_random(n, 7)
    for lower in (True, False):
        assert np.array_equal(O.Op.csc_sym(n, A.indptr, A.indices, A.data, lower).perform_op(x),
                              R.Op.csc_sym(n, A.indptr, A.indices, A.data, lower).perform_op(x))  # SparseSymMatProd.h:83-88
    assert np.array_equal(O.Op.csc(n, n, A.indptr, A.indices, A.data).perform_op(x), R.Op.csc(n, A.indptr, A.indices, A.data).perform_op(x))
    Ar = A.tocsr()
    Ar.sort_indices()
    assert np.array_equal(O.Op.csr(n, n, Ar.indptr, Ar.indices, Ar.data).perform_op(x), R.Op.csr(n, Ar.indptr, Ar.indices, Ar.data).perform_op(x))
    v0 = O.simple_random(n, 0)
    for symmetric, oop, rop in ((True, O.Op.csc_sym(n, A.indptr, A.indices, A.data, True), R.Op.csc_sym(n, A.indptr, A.indices, A.data, True)),
                                (False, O.Op.csc(n, n, A.indptr, A.indices, A.data), R.Op.csc(n, A.indptr, A.indices, A.data))):
        f = O.Factorization(oop, m, symmetric)
        f.init(v0)
        f.factorize_from(1, m)  # Lanczos.h:62-187 / Arnoldi.h:198-295
        V, H, ff = f.matrices()
        rV, rH, rf, beta, kk, nops = R.factorize(rop, m, v0, symmetric)
        assert np.array_equal(V, rV) and np.array_equal(H, rH) and np.array_equal(ff, rf)
        assert f.f_norm() == beta and f.subspace_dim() == kk and f.num_operations() == nops


@needs_ref
@pytest.mark.parametrize("n,prob,k,m", SPARSE_CASES)
@pytest.mark.parametrize("rule", RULES_SYM)
def test_symmetric_solves_equal_the_reference(n, prob, k, m, rule):
    A, S = sparse_fixture(n, prob)
    c, ev, U = oracle_sym(O.Op.csc_sym(n, A.indptr, A.indices, A.data, True), k, m, rule)
    r = R.symeigs(R.Op.csc_sym(n, A.indptr, A.indices, A.data, True), k, m, selection=getattr(R, rule))
    assert c == counters(r)  # nconv, info, num_iterations, num_operations: equal
    assert np.abs(ev - r.eigenvalues).max() <= 1e-12
    assert np.array_equal(ev, r.eigenvalues) and np.array_equal(U, r.eigenvectors)
    # and what the library returns now is what was committed
    assert counters(r) == list(GOLD[f"sym_{n}_{rule}"]) and np.array_equal(r.eigenvalues, GOLD[f"sym_{n}_{rule}_evals"])
    # the reference's own acceptance bar on its own code (test/SymEigs.cpp:62-75)
    assert r.info == 0 and np.abs(S @ r.eigenvectors - r.eigenvectors * r.eigenvalues).max() < 1e-9


@needs_ref
@pytest.mark.parametrize("n,prob,k,m", SPARSE_CASES)
@pytest.mark.parametrize("rule", RULES_GEN)
def test_general_solves_equal_the_reference(n, prob, k, m, rule):
    A, S = sparse_fixture(n, prob)
    c, ev, U = oracle_gen(O.Op.csc(n, n, A.indptr, A.indices, A.data), k, m, rule)
    r = R.geneigs(R.Op.csc(n, A.indptr, A.indices, A.data), k, m, selection=getattr(R, rule))
    assert c == counters(r)
    assert np.array_equal(ev, r.eigenvalues) and np.array_equal(U, r.eigenvectors)
    assert counters(r) == list(GOLD[f"gen_{n}_{rule}"])
    if r.info == 0:  # the non-converging rules (SmallestMagn on these fixtures) are compared too, but carry no residual claim
        assert np.abs(A @ r.eigenvectors - r.eigenvectors * r.eigenvalues).max() < 1e-9  # test/GenEigs.cpp:62-75


@needs_ref
def test_examples_and_edge_cases_equal_the_reference():
    M = cycle_laplacian(20)  # test/Example1.cpp: double eigenvalues, tol 1e-15
    for k, m in [(3, 6), (5, 12), (6, 12)]:
        c, ev, U = oracle_sym(O.Op.dense_sym(M), k, m, "LargestMagn", tol=1e-15, sorting=O.SmallestAlge)
        r = R.symeigs(R.Op.dense_sym(M), k, m, selection=R.LargestMagn, tol=1e-15, sorting=R.SmallestAlge)
        assert c == counters(r) and np.array_equal(ev, r.eigenvalues) and np.array_equal(U, r.eigenvectors)
    for M2 in EXAMPLE2:  # test/Example2.cpp: near rank one
        c, ev, U = oracle_sym(O.Op.dense_sym(M2), 1, 3, "LargestAlge")
        r = R.symeigs(R.Op.dense_sym(M2), 1, 3, selection=R.LargestAlge)
        assert c == counters(r) and np.array_equal(ev, r.eigenvalues) and np.array_equal(U, r.eigenvectors)
    # test/Example4.cpp:59-92: the zero matrix (expand_basis, Arnoldi.h:66-115) and a start vector in the null space (:162-169)
    rng = np.random.default_rng(123)
    n = 100
    Z = np.zeros((n, n))
    v0 = rng.uniform(-1, 1, n)
    c, ev, U = oracle_sym(O.Op.dense_sym(Z), 3, 6, "LargestAlge", v0=v0, sorting=O.SmallestAlge)
    r = R.symeigs(R.Op.dense_sym(Z), 3, 6, selection=R.LargestAlge, sorting=R.SmallestAlge, v0=v0)
    assert c == counters(r) and np.array_equal(ev, r.eigenvalues) and np.array_equal(U, r.eigenvectors)
    Mm = rng.uniform(-1, 1, (n, n))
    w, V = np.linalg.eigh(Mm + Mm.T)
    w[-1] = 0.0
    A = (V * w) @ V.T
    A = (A + A.T) / 2
    v0 = V[:, -1].copy()
    c, ev, U = oracle_sym(O.Op.dense_sym(A), 3, 6, "LargestAlge", v0=v0, sorting=O.SmallestAlge)
    r = R.symeigs(R.Op.dense_sym(A), 3, 6, selection=R.LargestAlge, sorting=R.SmallestAlge, v0=v0)
    assert c == counters(r) and np.array_equal(ev, r.eigenvalues) and np.array_equal(U, r.eigenvectors)
    # graded and clustered spectra (the beta < sqrt(eps) and iterative-correction paths of Lanczos.h:99-121, :156-180)
    for d in (np.logspace(-12, 0, 60), np.concatenate([1.0 + 1e-9 * np.arange(8), np.linspace(0, 0.5, 52)]), np.r_[np.zeros(30), np.ones(30)]):
        D = sp.diags(d).tocsc()
        for rule in ("LargestAlge", "SmallestAlge", "BothEnds"):
            c, ev, U = oracle_sym(O.Op.csc_sym(60, D.indptr, D.indices, D.data, True), 4, 12, rule, maxit=300)
            r = R.symeigs(R.Op.csc_sym(60, D.indptr, D.indices, D.data, True), 4, 12, selection=getattr(R, rule), maxit=300)
            assert c == counters(r) and np.array_equal(ev, r.eigenvalues) and np.array_equal(U, r.eigenvectors)


@needs_ref
def test_constructor_errors_are_the_reference_s():
    D = sp.diags(np.arange(1.0, 11.0)).tocsc()
    for nev, ncv in [(0, 5), (10, 11), (3, 3), (3, 11)]:  # HermEigsBase.h:267-271
        with pytest.raises(RuntimeError, match="must satisfy"):
            R.symeigs(R.Op.csc_sym(10, D.indptr, D.indices, D.data, True), nev, ncv)
        with pytest.raises(ValueError):
            O.SymEigsSolver(O.Op.diag(np.arange(1.0, 11.0)), nev, ncv)


_PROGRAM_RUNS = {}  # name -> future of (returncode, stdout): see _program_result


def _run_program(exe):
    import subprocess

    out = subprocess.run([exe], capture_output=True, text=True, timeout=1800)
    return out.returncode, out.stdout


def _program_result(name, directory):
    """Returncode and output of oracle/_ref/tests/<name>.bin.  In a serial pytest run the first call starts ALL the programs, one
    per core, and every test then collects its own — the eight slow ones (dense stand-in LU / complex algebra at n = 1000: 25-75 s
    each) would otherwise add seven minutes to the suite; under pytest-xdist every worker runs just the program it was asked for."""
    import concurrent.futures

    exe = os.path.join(directory, name + ".bin")
    if os.environ.get("PYTEST_XDIST_WORKER"):
        return _run_program(exe)
    if not _PROGRAM_RUNS:
        pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, os.cpu_count() or 1))
        for nm in R.REFERENCE_TEST_PROGRAMS:
            e = os.path.join(directory, nm + ".bin")
            if os.path.exists(e):
                _PROGRAM_RUNS[nm] = pool.submit(_run_program, e)
        pool.shutdown(wait=False)
    return _PROGRAM_RUNS[name].result()


@needs_ref
@pytest.mark.parametrize("name", R.REFERENCE_TEST_PROGRAMS)
def test_reference_s_own_test_programs_pass_on_the_stand_in_algebra(name):
    # /root/reference/test/<name>.cpp, unmodified, against /root/reference/include + oracle/eigen_shim: the reference's own acceptance
    # tests of its own code.  What this checks is the stand-in (an expression it mis-evaluated would fail the reference's bars).
    d = R.build_tests()
    exe = os.path.join(d or "", name + ".bin")
    if not d or not os.path.exists(exe):
        pytest.skip("oracle/_ref/tests not built")
    returncode, stdout = _program_result(name, d)
    # (test/RitzPairs.cpp has two test cases without assertions: Catch2 then reports "test cases: 2 | 2 passed")
    assert returncode == 0 and ("All tests passed" in stdout or " passed" in stdout) and "failed" not in stdout, stdout[-2000:]


def test_stand_in_decompositions_self_check(tmp_path):
    # oracle/eigen_shim's HouseholderQR / EigenSolver / ComplexEigenSolver (used, but only timed or compared in modulus, by the
    # reference's test/QR.cpp and test/Eigen.cpp): residuals of the stand-ins themselves.  Needs no reference.
    import subprocess

    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "shim_selfcheck")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-w", "-I" + os.path.join(here, "oracle", "eigen_shim"),
                           os.path.join(here, "oracle", "eigen_shim_selfcheck.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout


# ---- shift-and-invert drivers (SymEigsShiftSolver.h:190-215, GenEigsRealShiftSolver.h:52-58): the reference's back-transformation
# ---- and sorting of the Ritz values on its own code, the shift solve handed to both sides as the same callback


@needs_ref
@pytest.mark.parametrize("n,prob,k,m,sigma", SHIFT_SYM)
@pytest.mark.parametrize("rule", RULES_SYM)
def test_symmetric_shift_solves_equal_the_reference(n, prob, k, m, sigma, rule):
    import scipy.sparse.linalg as spla

    A, S = sparse_fixture(n, prob)
    lu = spla.splu((S - sigma * sp.identity(n)).tocsc())
    o = O.SymEigsSolver(O.Op.callback(n, lu.solve), k, m, sigma=sigma)
    o.init()
    nconv = o.compute(getattr(O, rule))
    r = R.symeigs_shift(R.Op.callback(n, lu.solve), k, m, sigma, selection=getattr(R, rule))
    assert [nconv, o.info(), o.num_iterations(), o.num_operations()] == counters(r)
    assert np.array_equal(o.eigenvalues(), r.eigenvalues) and np.array_equal(o.eigenvectors(), r.eigenvectors)
    if r.info == 0:  # the reference's own bar (test/SymEigsShift.cpp:62-75); SmallestMagn is allowed to fail there
        assert np.abs(S @ r.eigenvectors - r.eigenvectors * r.eigenvalues).max() < 1e-9


@needs_ref
@pytest.mark.parametrize("n,prob,k,m,sigma", SHIFT_GEN)
@pytest.mark.parametrize("rule", ["LargestMagn", "LargestReal", "LargestImag", "SmallestReal"])
def test_general_real_shift_solves_equal_the_reference(n, prob, k, m, sigma, rule):
    import scipy.sparse.linalg as spla

    A, S = sparse_fixture(n, prob)
    lu = spla.splu((A - sigma * sp.identity(n)).tocsc())
    o = O.GenEigsSolver(O.Op.callback(n, lu.solve), k, m, sigma=sigma)
    o.init()
    nconv = o.compute(getattr(O, rule))
    r = R.geneigs_real_shift(R.Op.callback(n, lu.solve), k, m, sigma, selection=getattr(R, rule))
    assert [nconv, o.info(), o.num_iterations(), o.num_operations()] == counters(r)
    assert np.array_equal(o.eigenvalues(), r.eigenvalues) and np.array_equal(o.eigenvectors(), r.eigenvectors)
    if r.info == 0:
        # test/GenEigsRealShift.cpp:58-70 asks 1e-9 with Eigen::SparseLU behind the operator; with scipy's LU behind the SAME
        # reference code the 1000 x 1000 fixture lands at 1.08e-9 — the claim here is the bit-equality above, not the solver's bar
        assert np.abs(A @ r.eigenvectors - r.eigenvectors * r.eigenvalues).max() < 1e-8


# ---- generalized drivers with B-inner products (SymGEigsSolver.h:224-238 regular inverse; SymGEigsShiftSolver.h:36-207 shift-invert /
# ---- buckling / Cayley; ArnoldiOp<Op, BOp>): the reference's code with B x from its own SparseSymMatProd and the inverses handed in
# ---- as callbacks; the oracle's Krylov callback is composed of the same pieces (its B product is bit-identical, tested above)
def oracle_b_solver(krylov, bop, n, k, m, transform, sigma):
    s = O.SymEigsSolver.__new__(O.SymEigsSolver)
    s._op, s._bop = O.Op.callback(n, krylov), bop
    s.h = O.lib().oracle_symeigs_create_b(s._op.h, s._bop.h, k, m, transform, float(sigma))
    assert s.h
    s.op, s.nev, s.ncv, s.n = s._op, k, min(m, n), n
    return s


def pencil_fixture(n, prob):
    """gen_sparse_data(n, A, B, prob) of test/SymGEigsRegInv.cpp:35-44: A = sprand (lower triangle used), B = A'A + 0.1 I."""
    r, c, v = O.gen_sparse_data(n, prob)
    A = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsc()
    A.sort_indices()
    B = (A.T @ A + 0.1 * sp.identity(n)).tocsc()
    B.sort_indices()
    low = lambda M: sp.tril(M).tocsc()
    return low(A), low(B)


@needs_ref
@pytest.mark.parametrize("n,prob,k,m", [(10, 0.5, 3, 6), (100, 0.1, 10, 20), (1000, 0.01, 20, 50)])  # test/SymGEigsRegInv.cpp:109-145
@pytest.mark.parametrize("rule", ["LargestMagn", "LargestAlge", "SmallestAlge", "BothEnds"])
def test_regular_inverse_driver_equals_the_reference(n, prob, k, m, rule):
    import scipy.sparse.linalg as spla

    Al, Bl = pencil_fixture(n, prob)
    sym = lambda L: (L + sp.tril(L, -1).T).tocsc()
    lu = spla.splu(sym(Bl))
    oa = O.Op.csc_sym(n, Al.indptr, Al.indices, Al.data, True)
    ob = O.Op.csc_sym(n, Bl.indptr, Bl.indices, Bl.data, True)
    o = oracle_b_solver(lambda x: lu.solve(oa.perform_op(x)), ob, n, k, m, 0, 0.0)
    o.init()
    nconv = o.compute(getattr(O, rule), 100)
    r = R.symgeigs_reginv(R.Op.csc_sym(n, Al.indptr, Al.indices, Al.data, True), R.Op.csc_sym(n, Bl.indptr, Bl.indices, Bl.data, True),
                          lu.solve, k, m, selection=getattr(R, rule), maxit=100)
    assert [nconv, o.info(), o.num_iterations(), o.num_operations()] == counters(r)
    assert np.array_equal(o.eigenvalues(), r.eigenvalues) and np.array_equal(o.eigenvectors(), r.eigenvectors)
    if r.info == 0:
        A, B = sym(Al), sym(Bl)
        assert np.abs(A @ r.eigenvectors - (B @ r.eigenvectors) * r.eigenvalues).max() < 1e-8


@needs_ref
@pytest.mark.parametrize("mode,transform", [("ShiftInvert", 1), ("Buckling", 2), ("Cayley", 3)])
@pytest.mark.parametrize("rule", ["LargestMagn", "LargestAlge", "SmallestAlge", "BothEnds"])
def test_generalized_shift_drivers_equal_the_reference(mode, transform, rule):
    import scipy.sparse.linalg as spla

    n, k, m, sigma = 100, 10, 20, 1.2345  # test/SymGEigsShift.cpp:121-141, :214-234, :307-327
    Al, Bl = pencil_fixture(n, 0.1)
    if mode == "Buckling":  # K = KG'KG + 0.1 I is the first operand, KG the second
        Al, Bl = Bl, Al
    sym = lambda L: (L + sp.tril(L, -1).T).tocsc()
    lu = spla.splu((sym(Al) - sigma * sym(Bl)).tocsc())
    Pl = Al if mode == "Buckling" else Bl  # the matrix of the inner product and of the product inside the operator
    op_p = O.Op.csc_sym(n, Pl.indptr, Pl.indices, Pl.data, True)
    if mode == "Cayley":
        krylov = lambda x: x + (2.0 * sigma) * lu.solve(op_p.perform_op(x))  # SymGEigsCayleyOp.h:91-100
    else:
        krylov = lambda x: lu.solve(op_p.perform_op(x))                      # SymGEigsShiftInvertOp.h / SymGEigsBucklingOp.h
    o = oracle_b_solver(krylov, O.Op.csc_sym(n, Pl.indptr, Pl.indices, Pl.data, True), n, k, m, transform, sigma)
    o.init()
    nconv = o.compute(getattr(O, rule), 100)
    r = R.symgeigs_shift(R.Op.callback(n, lu.solve), R.Op.csc_sym(n, Pl.indptr, Pl.indices, Pl.data, True), mode, k, m, sigma,
                         selection=getattr(R, rule), maxit=100)
    assert [nconv, o.info(), o.num_iterations(), o.num_operations()] == counters(r)
    assert np.array_equal(o.eigenvalues(), r.eigenvalues) and np.array_equal(o.eigenvectors(), r.eigenvectors)
    if r.info == 0:
        A, B = sym(Al), sym(Bl)
        assert np.abs(A @ r.eigenvectors - (B @ r.eigenvectors) * r.eigenvalues).max() < 1e-8


# ---- contrib/PartialSVDSolver.h:112-209: the reference's product operators (A'A for a tall matrix, AA' for a wide one) and its
# ---- driver on its own code; the oracle's restatement (oracle.partial_svd) with the two products taken from the oracle's operators
@needs_ref
@pytest.mark.parametrize("m,n,prob,ncomp,ncv", [(100, 20, 0.1, 5, 10), (20, 100, 0.1, 5, 10), (1000, 100, 0.01, 10, 30), (100, 1000, 0.01, 10, 30)])
def test_partial_svd_equals_the_reference(m, n, prob, ncomp, ncv):
    # test/SVD.cpp:17-33, :100-137 (sparse cases; the shapes of its tall and wide fixtures)
    r, c, v = O.gen_sparse_data_rect(m, n, prob)
    A = sp.coo_matrix((v, (r, c)), shape=(m, n)).tocsc()
    A.sort_indices()
    a_op = O.Op.csc(m, n, A.indptr, A.indices, A.data)                 # y = A x   (SparseGenMatProd's loop)
    at_op = O.Op.csr(n, m, A.indptr, A.indices, A.data)                # y = A' x: the same arrays read as the CSR rows of A'
    tall = m > n
    krylov = (lambda x: at_op.perform_op(a_op.perform_op(x))) if tall else (lambda x: a_op.perform_op(at_op.perform_op(x)))
    o = O.SymEigsSolver(O.Op.callback(min(m, n), krylov), ncomp, ncv)
    o.init()
    nconv = o.compute(O.LargestAlge, 1000, 1e-10)
    rn, sv, X = R.partial_svd(A, ncomp, ncv)
    assert nconv == rn == ncomp
    assert np.array_equal(np.sqrt(o.eigenvalues()), sv) and np.array_equal(o.eigenvectors(), X)
    dense = np.linalg.svd(A.toarray(), compute_uv=False)[:ncomp]
    assert np.abs(sv - dense).max() < 1e-9                               # test/SVD.cpp:53-60
    # and the restatement the GPU tests use (scipy products inside: another summation order) agrees to rounding
    n2, sv2, U2, V2 = O.partial_svd(A, ncomp, ncv)
    assert n2 == ncomp and np.abs(sv2 - sv).max() < 1e-12
