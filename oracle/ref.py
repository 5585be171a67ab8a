"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

ctypes face of ``oracle/_ref/libspectra_ref.so``: the REFERENCE'S OWN solver code
(yixuan/spectra headers compiled where they lie by ``oracle/build_ref.sh``, with
``oracle/eigen_shim`` standing in for Eigen).  Only ``tests/``, ``bench.py``'s
``cpu_baseline`` leg and ``__graft_entry__`` may import this module.

The library is built in the container that has ``/root/reference``; on the GPU box the
prebuilt file travels with the snapshot (``/root/reference`` does not exist there and is
never read at run time).  ``available()`` says whether the library can be loaded.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_ref", "libspectra_ref.so")
REFERENCE_DIR = os.environ.get("MISPEC_REFERENCE_DIR", "/root/reference")

# Util/SelectionRule.h:33-58 (same enumerator order as the oracle's)
LargestMagn, LargestReal, LargestImag, LargestAlge, SmallestMagn, SmallestReal, SmallestImag, SmallestAlge, BothEnds = range(9)

_CB = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double))


class _RefOp(C.Structure):
    _fields_ = [
        ("kind", C.c_int),
        ("n", C.c_long),
        ("ptr", C.POINTER(C.c_int)),
        ("ind", C.POINTER(C.c_int)),
        ("val", C.POINTER(C.c_double)),
        ("lower", C.c_int),
        ("cb", _CB),
    ]


def build(force=False):
    """(Re)build the library when the reference is present; returns the path or None."""
    if os.path.isdir(os.path.join(REFERENCE_DIR, "include", "Spectra")):
        subprocess.check_call([os.path.join(_HERE, "build_ref.sh")] + (["--force"] if force else []), stdout=subprocess.DEVNULL)
    return _LIB_PATH if os.path.exists(_LIB_PATH) else None


def available():
    return build() is not None


REFERENCE_TEST_PROGRAMS = ["SymEigs", "GenEigs", "Schur", "Example1", "Example2", "Example4", "SparseSymMatProd", "SparseGenMatProd",
                           "DenseSymMatProd", "DenseGenMatProd", "SymEigsShift", "GenEigsRealShift", "SymGEigsRegInv", "SymGEigsCholesky",
                           "Example3", "SVD", "Givens", "QR", "Eigen", "Arnoldi",
                           # the whole of test/CMakeLists.txt since the end of round 4 (complex solvers, BKLDLT, the Davidson family,
                           # the complex-shift solver, the dense / sparse pencils of SymGEigsShift)
                           "HermEigs", "ComplexEigs", "BKLDLT", "Orthogonalization", "JDSymEigsBase", "JDSymEigsDPRConstructor",
                           "RitzPairs", "SearchSpace", "DavidsonSymEigs", "GenEigsComplexShift", "SymGEigsShift"]


def build_tests():
    """The reference's own Catch2 programs on the reference's own headers + oracle/eigen_shim (oracle/_ref/tests/*.bin): built
    where the reference is present; returns the directory (the binaries may be prebuilt) or None."""
    if os.path.isdir(os.path.join(REFERENCE_DIR, "include", "Spectra")):
        subprocess.check_call([os.path.join(_HERE, "build_ref.sh"), "--tests"], stdout=subprocess.DEVNULL)
    d = os.path.join(_HERE, "_ref", "tests")
    return d if os.path.isdir(d) else None


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = build()
        if path is None:
            raise RuntimeError("oracle/_ref/libspectra_ref.so is not built and %s is absent" % REFERENCE_DIR)
        L = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        lp = C.POINTER(C.c_long)
        op = C.POINTER(_RefOp)
        sig = {
            "ref_last_error": (C.c_char_p, []),
            "ref_describe": (C.c_char_p, []),
            "ref_simple_random": (None, [C.c_ulong, C.c_long, dp]),
            "ref_givens": (None, [C.c_double, C.c_double, dp, dp, dp]),
            "ref_argsort": (C.c_int, [C.c_int, dp, C.c_long, lp]),
            "ref_tridiag_qr": (C.c_int, [C.c_long, dp, C.c_double, dp, dp, dp]),
            "ref_hess_qr": (C.c_int, [C.c_long, dp, C.c_double, dp, dp]),
            "ref_double_shift_qr": (C.c_int, [C.c_long, dp, C.c_double, C.c_double, dp, dp]),
            "ref_tridiag_eigen": (C.c_int, [C.c_long, dp, dp, dp]),
            "ref_hess_eigen": (C.c_int, [C.c_long, dp, dp, dp]),
            "ref_op_apply": (C.c_int, [op, dp, dp]),
            "ref_symeigs": (C.c_long, [op, C.c_long, C.c_long, dp, C.c_int, C.c_long, C.c_double, C.c_int, lp, dp, dp]),
            "ref_geneigs": (C.c_long, [op, C.c_long, C.c_long, dp, C.c_int, C.c_long, C.c_double, C.c_int, lp, dp, dp]),
            "ref_symeigs_shift": (C.c_long, [op, C.c_long, C.c_long, C.c_double, dp, C.c_int, C.c_long, C.c_double, C.c_int, lp, dp, dp]),
            "ref_geneigs_real_shift": (C.c_long, [op, C.c_long, C.c_long, C.c_double, dp, C.c_int, C.c_long, C.c_double, C.c_int, lp, dp, dp]),
            "ref_symgeigs_reginv": (C.c_long, [op, op, _CB, C.c_long, C.c_long, C.c_int, C.c_long, C.c_double, C.c_int, lp, dp, dp]),
            "ref_symgeigs_shift": (C.c_long, [op, op, C.c_int, C.c_long, C.c_long, C.c_double, C.c_int, C.c_long, C.c_double, C.c_int, lp, dp, dp]),
            "ref_partial_svd": (C.c_long, [C.c_long, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int), dp, C.c_long, C.c_long, C.c_long, C.c_double,
                                           lp, dp, dp]),
            "ref_factorize": (C.c_int, [op, C.c_long, C.c_int, dp, dp, dp, dp, dp]),
            "ref_symeigs_time": (C.c_double, [op, C.c_long, C.c_long, C.c_long, C.c_double, lp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _check(rc):
    if rc < 0:
        raise RuntimeError(lib().ref_last_error().decode())
    return rc


class Op:
    """An operator of the reference: SparseSymMatProd / SparseGenMatProd / DenseSymMatProd / DenseGenMatProd / a callback."""

    def __init__(self, kind, n, ptr=None, ind=None, val=None, lower=1, cb=None):
        self._keep = (
            None if ptr is None else np.ascontiguousarray(ptr, dtype=np.int32),
            None if ind is None else np.ascontiguousarray(ind, dtype=np.int32),
            None if val is None else np.ascontiguousarray(val, dtype=np.float64),
        )
        self.n = int(n)
        self._cb = _CB(cb) if cb is not None else _CB()
        self.c = _RefOp(
            kind, self.n,
            _ip(self._keep[0]) if ptr is not None else None,
            _ip(self._keep[1]) if ind is not None else None,
            _dp(self._keep[2]) if val is not None else None,
            int(lower), self._cb,
        )

    @classmethod
    def csc_sym(cls, n, colptr, rowind, val, lower=True):
        return cls(0, n, colptr, rowind, val, 1 if lower else 0)

    @classmethod
    def csc(cls, n, colptr, rowind, val):
        return cls(1, n, colptr, rowind, val)

    @classmethod
    def csr(cls, n, rowptr, colind, val):
        return cls(2, n, rowptr, colind, val)

    @classmethod
    def dense_sym(cls, A):
        A = np.asfortranarray(A, dtype=np.float64)
        return cls(3, A.shape[0], val=A.ravel(order="F"))

    @classmethod
    def dense_gen(cls, A):
        A = np.asfortranarray(A, dtype=np.float64)
        return cls(4, A.shape[0], val=A.ravel(order="F"))

    @classmethod
    def callback(cls, n, fn):
        def tramp(xp, yp):
            x = np.ctypeslib.as_array(xp, shape=(n,))
            y = np.ctypeslib.as_array(yp, shape=(n,))
            y[:] = fn(x)

        return cls(5, n, cb=tramp)

    def perform_op(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.n)
        _check(lib().ref_op_apply(C.byref(self.c), _dp(x), _dp(y)))
        return y


def simple_random(n, seed=0):
    out = np.empty(n)
    lib().ref_simple_random(seed, n, _dp(out))
    return out


def givens(x, y):
    r, c, s = C.c_double(), C.c_double(), C.c_double()
    lib().ref_givens(x, y, C.byref(r), C.byref(c), C.byref(s))
    return r.value, c.value, s.value


def argsort(rule, values):
    v = np.ascontiguousarray(values, dtype=np.float64)
    out = np.empty(len(v), dtype=np.int64)
    _check(lib().ref_argsort(rule, _dp(v), len(v), out.ctypes.data_as(C.POINTER(C.c_long))))
    return out


def tridiag_qr(T, shift):
    T = np.asfortranarray(T, dtype=np.float64)
    n = T.shape[0]
    R, H, Q = (np.empty((n, n), order="F") for _ in range(3))
    _check(lib().ref_tridiag_qr(n, _dp(T), shift, _dp(R), _dp(H), _dp(Q)))
    return R, H, Q


def hess_qr(H, shift):
    H = np.asfortranarray(H, dtype=np.float64)
    n = H.shape[0]
    Q, out = (np.empty((n, n), order="F") for _ in range(2))
    _check(lib().ref_hess_qr(n, _dp(H), shift, _dp(Q), _dp(out)))
    return Q, out


def double_shift_qr(H, s, t):
    H = np.asfortranarray(H, dtype=np.float64)
    n = H.shape[0]
    Q, out = (np.empty((n, n), order="F") for _ in range(2))
    _check(lib().ref_double_shift_qr(n, _dp(H), s, t, _dp(Q), _dp(out)))
    return Q, out


def tridiag_eigen(T):
    T = np.asfortranarray(T, dtype=np.float64)
    n = T.shape[0]
    ev = np.empty(n)
    U = np.empty((n, n), order="F")
    _check(lib().ref_tridiag_eigen(n, _dp(T), _dp(ev), _dp(U)))
    return ev, U


def hess_eigen(H):
    H = np.asfortranarray(H, dtype=np.float64)
    n = H.shape[0]
    ev = np.empty(2 * n)
    U = np.empty(2 * n * n)
    _check(lib().ref_hess_eigen(n, _dp(H), _dp(ev), _dp(U)))
    return ev.view(np.complex128), U.view(np.complex128).reshape((n, n), order="F")


def factorize(op, m, v0, symmetric=True):
    """Lanczos / Arnoldi of the reference: init(v0), factorize_from(1, m) -> V, H, f, beta, k, nops."""
    n = op.n
    v0 = np.ascontiguousarray(v0, dtype=np.float64)
    V = np.empty((n, m), order="F")
    H = np.empty((m, m), order="F")
    f = np.empty(n)
    scal = np.empty(3)
    _check(lib().ref_factorize(C.byref(op.c), m, 1 if symmetric else 0, _dp(v0), _dp(V), _dp(H), _dp(f), _dp(scal)))
    return V, H, f, scal[0], int(scal[1]), int(scal[2])


class Result:
    pass


def symeigs(op, nev, ncv, selection=LargestMagn, maxit=1000, tol=1e-10, sorting=LargestAlge, v0=None, vectors=True):
    """SymEigsSolver<Op>(op, nev, ncv); init(); compute(...) of the reference."""
    n = op.n
    counters = np.zeros(4, dtype=np.int64)
    evals = np.empty(nev)
    evecs = np.empty((n, nev), order="F") if vectors else None
    v0a = None if v0 is None else np.ascontiguousarray(v0, dtype=np.float64)
    k = _check(lib().ref_symeigs(C.byref(op.c), nev, ncv, None if v0a is None else _dp(v0a), selection, maxit, tol, sorting,
                                 counters.ctypes.data_as(C.POINTER(C.c_long)), _dp(evals), None if evecs is None else _dp(evecs)))
    r = Result()
    r.nconv, r.num_iterations, r.num_operations, r.info = (int(c) for c in counters)
    r.eigenvalues = evals[:k].copy()
    r.eigenvectors = None if evecs is None else evecs[:, :k].copy()
    return r


def geneigs(op, nev, ncv, selection=LargestMagn, maxit=1000, tol=1e-10, sorting=LargestMagn, v0=None, vectors=True):
    """GenEigsSolver<Op>(op, nev, ncv); init(); compute(...) of the reference."""
    n = op.n
    counters = np.zeros(4, dtype=np.int64)
    evals = np.empty(2 * nev)
    evecs = np.empty(2 * n * nev) if vectors else None
    v0a = None if v0 is None else np.ascontiguousarray(v0, dtype=np.float64)
    k = _check(lib().ref_geneigs(C.byref(op.c), nev, ncv, None if v0a is None else _dp(v0a), selection, maxit, tol, sorting,
                                 counters.ctypes.data_as(C.POINTER(C.c_long)), _dp(evals), None if evecs is None else _dp(evecs)))
    r = Result()
    r.nconv, r.num_iterations, r.num_operations, r.info = (int(c) for c in counters)
    r.eigenvalues = evals.view(np.complex128)[:k].copy()
    r.eigenvectors = None if evecs is None else evecs.view(np.complex128).reshape((n, nev), order="F")[:, :k].copy()
    return r


def symeigs_shift(op, nev, ncv, sigma, selection=LargestMagn, maxit=1000, tol=1e-10, sorting=LargestAlge, v0=None, vectors=True):
    """SymEigsShiftSolver<Op>(op, nev, ncv, sigma) of the reference; `op` is a callback operator that applies
    (A - sigma I)^{-1} (the reference's SparseSymShiftSolve delegates to Eigen::SparseLU, which the stand-in does not have)."""
    n = op.n
    counters = np.zeros(4, dtype=np.int64)
    evals = np.empty(nev)
    evecs = np.empty((n, nev), order="F") if vectors else None
    v0a = None if v0 is None else np.ascontiguousarray(v0, dtype=np.float64)
    k = _check(lib().ref_symeigs_shift(C.byref(op.c), nev, ncv, float(sigma), None if v0a is None else _dp(v0a), selection, maxit, tol, sorting,
                                       counters.ctypes.data_as(C.POINTER(C.c_long)), _dp(evals), None if evecs is None else _dp(evecs)))
    r = Result()
    r.nconv, r.num_iterations, r.num_operations, r.info = (int(c) for c in counters)
    r.eigenvalues = evals[:k].copy()
    r.eigenvectors = None if evecs is None else evecs[:, :k].copy()
    return r


def geneigs_real_shift(op, nev, ncv, sigma, selection=LargestMagn, maxit=1000, tol=1e-10, sorting=LargestMagn, v0=None, vectors=True):
    """GenEigsRealShiftSolver<Op>(op, nev, ncv, sigma) of the reference; `op` applies (A - sigma I)^{-1}."""
    n = op.n
    counters = np.zeros(4, dtype=np.int64)
    evals = np.empty(2 * nev)
    evecs = np.empty(2 * n * nev) if vectors else None
    v0a = None if v0 is None else np.ascontiguousarray(v0, dtype=np.float64)
    k = _check(lib().ref_geneigs_real_shift(C.byref(op.c), nev, ncv, float(sigma), None if v0a is None else _dp(v0a), selection, maxit, tol,
                                            sorting, counters.ctypes.data_as(C.POINTER(C.c_long)), _dp(evals),
                                            None if evecs is None else _dp(evecs)))
    r = Result()
    r.nconv, r.num_iterations, r.num_operations, r.info = (int(c) for c in counters)
    r.eigenvalues = evals.view(np.complex128)[:k].copy()
    r.eigenvectors = None if evecs is None else evecs.view(np.complex128).reshape((n, nev), order="F")[:, :k].copy()
    return r


def _sym_result(k, counters, evals, evecs):
    r = Result()
    r.nconv, r.num_iterations, r.num_operations, r.info = (int(c) for c in counters)
    r.eigenvalues = evals[:k].copy()
    r.eigenvectors = None if evecs is None else evecs[:, :k].copy()
    return r


def symgeigs_reginv(a, b, bsolve, nev, ncv, selection=LargestMagn, maxit=1000, tol=1e-10, sorting=LargestAlge):
    """SymGEigsSolver<SparseSymMatProd, BOp, GEigsMode::RegularInverse>(A, B, nev, ncv) of the reference; a, b: Op.csc_sym
    (lower triangles); bsolve(x) = B^{-1} x (the reference's SparseRegularInverse delegates that to Eigen::ConjugateGradient)."""
    n = a.n

    def tramp(xp, yp):
        x = np.ctypeslib.as_array(xp, shape=(n,))
        y = np.ctypeslib.as_array(yp, shape=(n,))
        y[:] = bsolve(x)

    cb = _CB(tramp)
    counters = np.zeros(4, dtype=np.int64)
    evals, evecs = np.empty(nev), np.empty((n, nev), order="F")
    k = _check(lib().ref_symgeigs_reginv(C.byref(a.c), C.byref(b.c), cb, nev, ncv, selection, maxit, tol, sorting,
                                         counters.ctypes.data_as(C.POINTER(C.c_long)), _dp(evals), _dp(evecs)))
    return _sym_result(k, counters, evals, evecs)


def symgeigs_shift(inv, b, mode, nev, ncv, sigma, selection=LargestMagn, maxit=1000, tol=1e-10, sorting=LargestAlge):
    """SymGEigsShiftSolver<Op, SparseSymMatProd, mode>(op, Bop, nev, ncv, sigma) of the reference; mode "ShiftInvert" /
    "Buckling" / "Cayley"; inv: Op.callback applying (A - sigma B)^{-1} (buckling: (K - sigma KG)^{-1}); b: Op.csc_sym of the
    matrix of the inner product (B; K for buckling)."""
    n = b.n
    counters = np.zeros(4, dtype=np.int64)
    evals, evecs = np.empty(nev), np.empty((n, nev), order="F")
    k = _check(lib().ref_symgeigs_shift(C.byref(inv.c), C.byref(b.c), {"ShiftInvert": 1, "Buckling": 2, "Cayley": 3}[mode], nev, ncv,
                                        float(sigma), selection, maxit, tol, sorting, counters.ctypes.data_as(C.POINTER(C.c_long)),
                                        _dp(evals), _dp(evecs)))
    return _sym_result(k, counters, evals, evecs)


def partial_svd(A, ncomp, ncv, maxit=1000, tol=1e-10):
    """contrib/PartialSVDSolver.h of the reference on a scipy sparse matrix (CSC): (nconv, singular values, X) with X the
    eigenvectors of the product operator — V for a tall matrix (rows > cols), U otherwise."""
    import scipy.sparse as sp

    A = sp.csc_matrix(A)
    A.sort_indices()
    m, n = A.shape
    cp, ri, v = np.ascontiguousarray(A.indptr, dtype=np.int32), np.ascontiguousarray(A.indices, dtype=np.int32), np.ascontiguousarray(A.data, dtype=np.float64)
    counters = np.zeros(4, dtype=np.int64)
    sv = np.empty(ncomp)
    dim = min(m, n)
    X = np.empty((dim, ncomp), order="F")
    k = _check(lib().ref_partial_svd(m, n, _ip(cp), _ip(ri), _dp(v), ncomp, ncv, maxit, tol, counters.ctypes.data_as(C.POINTER(C.c_long)),
                                     _dp(sv), _dp(X)))
    return int(counters[0]), sv[:k].copy(), X[:, :k].copy()


def symeigs_time(op, nev, ncv, maxit, tol=1e-10):
    """Wall seconds of init() + compute(maxit) of the reference's solver, and its counters."""
    counters = np.zeros(4, dtype=np.int64)
    t = lib().ref_symeigs_time(C.byref(op.c), nev, ncv, maxit, tol, counters.ctypes.data_as(C.POINTER(C.c_long)))
    if t < 0:
        raise RuntimeError(lib().ref_last_error().decode())
    return t, [int(c) for c in counters]
