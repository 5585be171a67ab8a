"""TEST INFRASTRUCTURE — ctypes binding of the CPU oracle (oracle/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (spectra_amd/, include/) never does.

The oracle is an Eigen-free C++ restatement of yixuan/spectra v1.2.0's
symmetric IRLM path; see oracle/spectra_oracle.hpp for the file:line map and
the pinning status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

# Util/SelectionRule.h:33-58 (same enumerator order)
LargestMagn, LargestReal, LargestImag, LargestAlge, SmallestMagn, SmallestReal, SmallestImag, SmallestAlge, BothEnds = range(9)
# Util/CompInfo.h:17-32
Successful, NotComputed, NotConverging, NumericalIssue = range(4)


def build(force=False):
    """Compile oracle/liboracle.so with the committed Makefile (g++ -O2)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("oracle_capi.cpp", "spectra_oracle.hpp", "spectra_oracle_gen.hpp", "synth_matrix.h", "onesweep_variant.hpp")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        lp = C.POINTER(C.c_long)
        vp = C.c_void_p
        sig = {
            "oracle_last_error": (C.c_char_p, []),
            "oracle_simple_random": (None, [C.c_ulong, C.c_long, dp]),
            "oracle_lcg_states": (None, [C.c_long, C.c_long, lp]),
            "oracle_minstd_states": (None, [C.c_long, C.c_long, lp]),
            "oracle_gen_sparse_data": (C.c_long, [C.c_int, C.c_double, ip, ip, dp]),
            "oracle_gen_sparse_data_rect": (C.c_long, [C.c_int, C.c_int, C.c_double, ip, ip, dp]),
            "oracle_gen_davidson_sparse": (C.c_long, [C.c_int, ip, ip, dp]),
            "oracle_synth_band_csr": (C.c_long, [C.c_long, C.c_ulonglong, lp, C.c_int, C.c_int, ip, ip, dp]),
            "oracle_synth_value": (C.c_double, [C.c_ulonglong, C.c_ulonglong, C.c_ulonglong]),
            "oracle_givens": (None, [C.c_double, C.c_double, dp, dp, dp]),
            "oracle_eigen_make_givens": (None, [C.c_double, C.c_double, dp, dp]),
            "oracle_tridiag_qr": (C.c_int, [C.c_long, dp, C.c_double, dp, dp, dp]),
            "oracle_tridiag_eigen": (C.c_int, [C.c_long, dp, dp, dp]),
            "oracle_argsort": (C.c_int, [C.c_int, dp, C.c_long, lp]),
            "oracle_op_csc_sym": (vp, [C.c_long, ip, ip, dp, C.c_int]),
            "oracle_op_csr": (vp, [C.c_long, C.c_long, ip, ip, dp]),
            "oracle_op_csc": (vp, [C.c_long, C.c_long, ip, ip, dp]),
            "oracle_op_dense_sym": (vp, [C.c_long, dp]),
            "oracle_op_dense_gen": (vp, [C.c_long, dp]),
            "oracle_op_diag": (vp, [C.c_long, dp]),
            "oracle_op_callback": (vp, [C.c_long, C.CFUNCTYPE(None, dp, dp)]),
            "oracle_op_free": (None, [vp]),
            "oracle_geigs_reginv_create": (vp, [C.c_long, ip, ip, dp, ip, ip, dp, C.c_long, C.c_long]),
            "oracle_geigs_free": (None, [vp]),
            "oracle_geigs_symeigs": (vp, [vp]),
            "oracle_geigs_cg_solve": (C.c_long, [vp, dp, dp]),
            "oracle_geigs_bprod": (None, [vp, dp, dp]),
            "oracle_op_rows": (C.c_long, [vp]),
            "oracle_op_apply": (None, [vp, dp, dp]),
            "oracle_op_time": (C.c_double, [vp, dp, dp, C.c_int]),
            "oracle_fac_create": (vp, [vp, C.c_long, C.c_int]),
            "oracle_fac_free": (None, [vp]),
            "oracle_fac_init": (C.c_int, [vp, dp]),
            "oracle_fac_factorize": (C.c_int, [vp, C.c_long, C.c_long]),
            "oracle_fac_k": (C.c_long, [vp]),
            "oracle_fac_nmatop": (C.c_long, [vp]),
            "oracle_fac_beta": (C.c_double, [vp]),
            "oracle_fac_get": (None, [vp, dp, dp, dp]),
            "oracle_symeigs_create": (vp, [vp, C.c_long, C.c_long]),
            "oracle_symeigs_free": (None, [vp]),
            "oracle_symeigs_set_shift_invert": (None, [vp, C.c_double]),
            "oracle_symeigs_set_onesweep": (None, [vp, C.c_int]),
            "oracle_symeigs_onesweep_stats": (None, [vp, dp]),
            "oracle_geneigs_set_shift_invert": (None, [vp, C.c_double]),
            "oracle_geneigs_set_complex_shift": (None, [vp, C.c_double, C.c_double, vp]),
            "oracle_complex_shift_probe": (C.c_double, [C.c_double]),
            "oracle_symeigs_create_b": (vp, [vp, vp, C.c_long, C.c_long, C.c_int, C.c_double]),
            "oracle_symeigs_init": (C.c_int, [vp, dp]),
            "oracle_symeigs_compute": (C.c_long, [vp, C.c_int, C.c_long, C.c_double, C.c_int]),
            "oracle_symeigs_info": (C.c_int, [vp]),
            "oracle_symeigs_num_iterations": (C.c_long, [vp]),
            "oracle_symeigs_num_operations": (C.c_long, [vp]),
            "oracle_symeigs_eigenvalues": (C.c_long, [vp, dp]),
            "oracle_symeigs_eigenvectors": (C.c_long, [vp, C.c_long, dp]),
            "oracle_symeigs_time_steps": (C.c_double, [vp, C.c_long, C.c_long, lp]),
            "oracle_symeigs_time_cycles": (C.c_int, [vp, C.c_long, C.c_long, C.c_int, C.c_double, C.c_long, dp]),
            "oracle_hess_qr": (C.c_int, [C.c_long, dp, C.c_double, dp, dp]),
            "oracle_double_shift_qr": (C.c_int, [C.c_long, dp, C.c_double, C.c_double, dp, dp]),
            "oracle_hess_schur": (C.c_int, [C.c_long, dp, dp, dp]),
            "oracle_hess_eigen": (C.c_int, [C.c_long, dp, dp, dp]),
            "oracle_geneigs_create": (vp, [vp, C.c_long, C.c_long]),
            "oracle_geneigs_free": (None, [vp]),
            "oracle_geneigs_init": (C.c_int, [vp, dp]),
            "oracle_geneigs_compute": (C.c_long, [vp, C.c_int, C.c_long, C.c_double, C.c_int]),
            "oracle_geneigs_info": (C.c_int, [vp]),
            "oracle_geneigs_num_iterations": (C.c_long, [vp]),
            "oracle_geneigs_num_operations": (C.c_long, [vp]),
            "oracle_geneigs_eigenvalues": (C.c_long, [vp, dp]),
            "oracle_geneigs_eigenvectors": (C.c_long, [vp, C.c_long, dp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_long))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _check(rc):
    if rc == -1:
        raise ValueError(lib().oracle_last_error().decode())  # std::invalid_argument
    if rc == -2:
        raise AssertionError(lib().oracle_last_error().decode())  # std::logic_error
    if rc < 0:
        raise RuntimeError(lib().oracle_last_error().decode())  # std::runtime_error


# ---------------------------------------------------------------------------------
def simple_random(n, seed=0):
    out = np.empty(n)
    lib().oracle_simple_random(seed, n, _dp(out))
    return out


def lcg_states(seed, count):
    out = np.empty(count, dtype=np.int64)
    lib().oracle_lcg_states(seed, count, _lp(out))
    return out


def minstd_states(seed, count):
    out = np.empty(count, dtype=np.int64)
    lib().oracle_minstd_states(seed, count, _lp(out))
    return out


def gen_sparse_data(n, prob):
    """test/SymEigs.cpp:25-42 fixture as COO (rows, cols, vals) — NOT symmetric."""
    cnt = lib().oracle_gen_sparse_data(n, prob, None, None, None)
    r = np.empty(cnt, dtype=np.int32)
    c = np.empty(cnt, dtype=np.int32)
    v = np.empty(cnt)
    lib().oracle_gen_sparse_data(n, prob, _ip(r), _ip(c), _dp(v))
    return r, c, v


def gen_sparse_data_rect(m, n, prob):
    """test/SVD.cpp:17-33 fixture (m x n) as COO (rows, cols, vals)."""
    cnt = lib().oracle_gen_sparse_data_rect(m, n, prob, None, None, None)
    r = np.empty(cnt, dtype=np.int32)
    c = np.empty(cnt, dtype=np.int32)
    v = np.empty(cnt)
    lib().oracle_gen_sparse_data_rect(m, n, prob, _ip(r), _ip(c), _dp(v))
    return r, c, v


def complex_shift_probe(sigmar):
    """The real probe shift GenEigsComplexShiftSolver draws from SimpleRandom(0) (GenEigsComplexShiftSolver.h:69-72)."""
    return float(lib().oracle_complex_shift_probe(float(sigmar)))


def gen_davidson_sparse(n):
    """test/DavidsonSymEigs.cpp:46-67 gen_sym_data_sparse(n) as COO (rows, cols, vals) — NOT symmetric."""
    cnt = lib().oracle_gen_davidson_sparse(n, None, None, None)
    r = np.empty(cnt, dtype=np.int32)
    c = np.empty(cnt, dtype=np.int32)
    v = np.empty(cnt)
    lib().oracle_gen_davidson_sparse(n, _ip(r), _ip(c), _dp(v))
    return r, c, v


def partial_svd(A, ncomp, ncv, maxit=1000, tol=1e-10):
    """contrib/PartialSVDSolver.h:112-209 restated on top of the SymEigs oracle: A is a scipy sparse matrix; the
    operator A'A (tall) / AA' (wide) is applied through a callback (two scipy products, :64-71 / :102-109).
    Returns (nconv, singular values, U, V)."""
    import scipy.sparse as sp

    A = sp.csr_matrix(A)
    At = sp.csr_matrix(A.T)
    m, n = A.shape
    tall = m > n
    dim = min(m, n)
    op = Op.callback(dim, (lambda x: At @ (A @ x)) if tall else (lambda x: A @ (At @ x)))
    eigs = SymEigsSolver(op, ncomp, ncv)
    eigs.init()
    nconv = eigs.compute(LargestAlge, maxit, tol)
    ev = eigs.eigenvalues()
    X = eigs.eigenvectors()
    sv = np.sqrt(ev)
    if tall:
        V = X
        U = A @ (X / sv)
    else:
        U = X
        V = At @ (X / sv)
    return nconv, sv, U, V


BAND_OFFSETS = (1, 2, 3, 1000, 1001, 100000, 100001)  # SURVEY §8(d) M-band
SYNTH_SEED = 20240607


def synth_band_csr(n, offsets=BAND_OFFSETS, seed=SYNTH_SEED, symmetric=True):
    offs = np.ascontiguousarray(offsets, dtype=np.int64)
    cnt = lib().oracle_synth_band_csr(n, seed, _lp(offs), len(offs), int(symmetric), None, None, None)
    rowptr = np.empty(n + 1, dtype=np.int32)
    colind = np.empty(cnt, dtype=np.int32)
    val = np.empty(cnt)
    lib().oracle_synth_band_csr(n, seed, _lp(offs), len(offs), int(symmetric), _ip(rowptr), _ip(colind), _dp(val))
    return rowptr, colind, val


def givens(x, y):
    r, c, s = C.c_double(), C.c_double(), C.c_double()
    lib().oracle_givens(x, y, C.byref(r), C.byref(c), C.byref(s))
    return r.value, c.value, s.value


def eigen_make_givens(p, q):
    c, s = C.c_double(), C.c_double()
    lib().oracle_eigen_make_givens(p, q, C.byref(c), C.byref(s))
    return c.value, s.value


def tridiag_qr(T, shift):
    """Returns (R, QtHQ, Q) as numpy (n, n) arrays."""
    T = np.asfortranarray(T, dtype=np.float64)
    n = T.shape[0]
    R, D, Q = (np.empty((n, n), order="F") for _ in range(3))
    _check(lib().oracle_tridiag_qr(n, _dp(T), shift, _dp(R), _dp(D), _dp(Q)))
    return R, D, Q


def tridiag_eigen(T):
    T = np.asfortranarray(T, dtype=np.float64)
    n = T.shape[0]
    ev = np.empty(n)
    U = np.empty((n, n), order="F")
    _check(lib().oracle_tridiag_eigen(n, _dp(T), _dp(ev), _dp(U)))
    return ev, U


def argsort(rule, values):
    v = _f64(values)
    out = np.empty(len(v), dtype=np.int64)
    _check(lib().oracle_argsort(rule, _dp(v), len(v), _lp(out)))
    return out


class Op:
    """A matrix operator of the oracle (the reference's OpType concept)."""

    def __init__(self, handle, n, keep=()):
        self.h = handle
        self.n = n
        self._keep = keep

    @classmethod
    def csc_sym(cls, n, colptr, rowind, val, lower=True):
        cp, ri, v = _i32(colptr), _i32(rowind), _f64(val)
        return cls(lib().oracle_op_csc_sym(n, _ip(cp), _ip(ri), _dp(v), int(lower)), n)

    @classmethod
    def csr(cls, nr, nc, rowptr, colind, val):
        rp, ci, v = _i32(rowptr), _i32(colind), _f64(val)
        return cls(lib().oracle_op_csr(nr, nc, _ip(rp), _ip(ci), _dp(v)), nr)

    @classmethod
    def csc(cls, nr, nc, colptr, rowind, val):
        cp, ri, v = _i32(colptr), _i32(rowind), _f64(val)
        return cls(lib().oracle_op_csc(nr, nc, _ip(cp), _ip(ri), _dp(v)), nr)

    @classmethod
    def dense_sym(cls, A):
        A = np.asfortranarray(A, dtype=np.float64)
        return cls(lib().oracle_op_dense_sym(A.shape[0], _dp(A)), A.shape[0])

    @classmethod
    def dense_gen(cls, A):
        A = np.asfortranarray(A, dtype=np.float64)
        return cls(lib().oracle_op_dense_gen(A.shape[0], _dp(A)), A.shape[0])

    @classmethod
    def callback(cls, n, fn):
        """fn(x: ndarray) -> ndarray; e.g. a scipy LU solve standing in for Eigen::SparseLU (SparseSymShiftSolve.h:104-109)."""
        def tramp(xp, yp):
            x = np.ctypeslib.as_array(xp, shape=(n,))
            np.ctypeslib.as_array(yp, shape=(n,))[:] = fn(x)
        cb = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double))(tramp)
        return cls(lib().oracle_op_callback(n, cb), n, keep=(cb,))

    @classmethod
    def diag(cls, d):
        d = _f64(d)
        return cls(lib().oracle_op_diag(len(d), _dp(d)), len(d))

    def rows(self):
        return self.n

    def cols(self):
        return self.n

    def perform_op(self, x):
        x = _f64(x)
        y = np.empty(self.n)
        lib().oracle_op_apply(self.h, _dp(x), _dp(y))
        return y

    def time_op(self, x, reps):
        x = _f64(x)
        y = np.empty(self.n)
        return lib().oracle_op_time(self.h, _dp(x), _dp(y), reps)

    def __del__(self):
        try:
            lib().oracle_op_free(self.h)
        except Exception:
            pass


class Factorization:
    """LinAlg/Arnoldi.h + LinAlg/Lanczos.h (symmetric=True follows Lanczos::factorize_from)."""

    def __init__(self, op, m, symmetric=True):
        self.op, self.m, self.n = op, m, op.n
        self.h = lib().oracle_fac_create(op.h, m, int(symmetric))

    def init(self, v0):
        v0 = _f64(v0)
        _check(lib().oracle_fac_init(self.h, _dp(v0)))

    def factorize_from(self, from_k, to_m):
        _check(lib().oracle_fac_factorize(self.h, from_k, to_m))

    def subspace_dim(self):
        return lib().oracle_fac_k(self.h)

    def num_operations(self):
        return lib().oracle_fac_nmatop(self.h)

    def f_norm(self):
        return lib().oracle_fac_beta(self.h)

    def matrices(self):
        V = np.empty((self.n, self.m), order="F")
        H = np.empty((self.m, self.m), order="F")
        f = np.empty(self.n)
        lib().oracle_fac_get(self.h, _dp(V), _dp(H), _dp(f))
        return V, H, f

    def __del__(self):
        try:
            lib().oracle_fac_free(self.h)
        except Exception:
            pass


class SymEigsSolver:
    """SymEigsSolver.h:133-160 (HermEigsBase.h) on the oracle; same method names as the reference."""

    def __init__(self, op, nev, ncv, sigma=None):
        self.op, self.nev, self.ncv, self.n = op, nev, min(ncv, op.n), op.n
        self.h = lib().oracle_symeigs_create(op.h, nev, ncv)
        if not self.h:
            raise ValueError(lib().oracle_last_error().decode())
        if sigma is not None:  # SymEigsShiftSolver.h:190-195 (the op must already be shift-inverted)
            lib().oracle_symeigs_set_shift_invert(self.h, sigma)

    def set_onesweep(self, on=True, fused=False, recorrect=False, one_reduction=False):
        """NOT the reference: switch the factorisation to the CPU restatement of this repository's opt-in one-sweep
        variant (oracle/onesweep_variant.hpp), for variant-vs-reference comparisons under the same driver.  fused: the last
        correction of every sweep rides on the restart (the device's default in that mode); recorrect: test hook, one more
        correction after every such restart; one_reduction: the product is applied to the un-normalised residual so that a step needs
        ONE reduction (its <f, Af> joins the previous pass's record; DESIGN.md 8 — CPU restatement only so far)."""
        lib().oracle_symeigs_set_onesweep(self.h, int(bool(on)) | (2 if fused else 0) | (4 if recorrect else 0) | (8 if one_reduction else 0))

    def onesweep_stats(self):
        out = np.zeros(10)
        lib().oracle_symeigs_onesweep_stats(self.h, _dp(out))
        keys = ("lagged_steps", "faithful_steps", "fallbacks_check", "fallbacks_state", "final_passes", "max_rel_c", "max_chk",
                "fused_restarts", "fused_recorrected", "one_reduction_steps")
        return dict(zip(keys, out.tolist()))

    def init(self, v0=None):
        v0 = None if v0 is None else _f64(v0)
        _check(lib().oracle_symeigs_init(self.h, _dp(v0)))

    def compute(self, selection=LargestMagn, maxit=1000, tol=1e-10, sorting=LargestAlge):
        rc = lib().oracle_symeigs_compute(self.h, selection, maxit, tol, sorting)
        _check(rc)
        return rc

    def info(self):
        return lib().oracle_symeigs_info(self.h)

    def num_iterations(self):
        return lib().oracle_symeigs_num_iterations(self.h)

    def num_operations(self):
        return lib().oracle_symeigs_num_operations(self.h)

    def eigenvalues(self):
        out = np.empty(self.nev)
        cnt = lib().oracle_symeigs_eigenvalues(self.h, _dp(out))
        return out[:cnt].copy()

    def eigenvectors(self, nvec=None):
        nvec = self.nev if nvec is None else nvec
        out = np.zeros((self.n, max(nvec, 1)), order="F")
        cnt = lib().oracle_symeigs_eigenvectors(self.h, nvec, _dp(out))
        return out[:, :cnt].copy(order="F")

    def __del__(self):
        try:
            lib().oracle_symeigs_free(self.h)
        except Exception:
            pass


class SymGEigsRegInvSolver(SymEigsSolver):
    """SymGEigsSolver<SparseSymMatProd, SparseRegularInverse, GEigsMode::RegularInverse> (SymGEigsSolver.h:224-238) on the
    oracle: A, B scipy sparse matrices of which the lower triangle is used; B^{-1} by the restated conjugate gradient,
    inner products in the B-inner product."""

    def __init__(self, A, B, nev, ncv):
        import scipy.sparse as sp

        A, B = sp.csc_matrix(A), sp.csc_matrix(B)
        A.sort_indices()
        B.sort_indices()
        n = A.shape[0]
        self._keep = (_i32(A.indptr), _i32(A.indices), _f64(A.data), _i32(B.indptr), _i32(B.indices), _f64(B.data))
        k = self._keep
        self.holder = lib().oracle_geigs_reginv_create(n, _ip(k[0]), _ip(k[1]), _dp(k[2]), _ip(k[3]), _ip(k[4]), _dp(k[5]), nev, ncv)
        if not self.holder:
            raise ValueError(lib().oracle_last_error().decode())
        self.h = lib().oracle_geigs_symeigs(self.holder)
        self.op, self.nev, self.ncv, self.n = None, nev, min(ncv, n), n

    def cg_solve(self, rhs):
        """B^{-1} rhs; returns (x, iterations) — iterations == -1: not converged."""
        x = np.empty(self.n)
        it = lib().oracle_geigs_cg_solve(self.holder, _dp(_f64(rhs)), _dp(x))
        return x, it

    def b_product(self, x):
        y = np.empty(self.n)
        lib().oracle_geigs_bprod(self.holder, _dp(_f64(x)), _dp(y))
        return y

    def __del__(self):
        try:
            lib().oracle_geigs_free(self.holder)
        except Exception:
            pass


class SymGEigsCholeskySolver(SymEigsSolver):
    """SymGEigsSolver<SparseSymMatProd, SparseCholesky, GEigsMode::Cholesky> (SymGEigsSolver.h:142-208) on the oracle:
    the standard problem L^{-1} A L^{-T} y = lambda y through a callback (dense Cholesky of B by numpy — the reference
    delegates it to Eigen::SimplicialLLT — and scipy triangular solves), eigenvectors x = L^{-T} y."""

    def __init__(self, A, B, nev, ncv):
        import scipy.linalg as sla
        import scipy.sparse as sp

        sym = lambda M: (sp.tril(M) + sp.tril(M, -1).T).tocsr()
        As = sym(sp.csc_matrix(A))
        self._L = np.linalg.cholesky(sym(sp.csc_matrix(B)).toarray())
        self._sla = sla
        L = self._L
        n = As.shape[0]
        fn = lambda x: sla.solve_triangular(L, As @ sla.solve_triangular(L, x, lower=True, trans="T"), lower=True)
        self._op = Op.callback(n, fn)
        super().__init__(self._op, nev, ncv)

    def eigenvectors(self, nvec=None):
        Y = super().eigenvectors(nvec)
        return self._sla.solve_triangular(self._L, Y, lower=True, trans="T")


class SymGEigsShiftSolver(SymEigsSolver):
    """SymGEigsShiftSolver<SymShiftInvert, SparseSymMatProd, mode> (SymGEigsShiftSolver.h:36-207) on the oracle.
    A, B: scipy sparse matrices whose lower triangles define the symmetric pencil (for mode "Buckling": A = K, B = KG and
    the inner product / product operator is K, as in the reference).  inv(A - sigma B) is a scipy sparse LU applied
    through a callback (the reference delegates it to Eigen::SparseLU, third party)."""

    MODES = {"ShiftInvert": 1, "Buckling": 2, "Cayley": 3}

    def __init__(self, A, B, nev, ncv, sigma, mode="ShiftInvert"):
        import scipy.sparse as sp
        import scipy.sparse.linalg as spla

        sym = lambda M: (sp.tril(M) + sp.tril(M, -1).T).tocsc()
        As, Bs = sym(sp.csc_matrix(A)), sym(sp.csc_matrix(B))
        n = As.shape[0]
        lu = spla.splu((As - sigma * Bs).tocsc())
        P = As if mode == "Buckling" else Bs  # the BOpType: K for buckling, B otherwise
        Plow = sp.tril(P).tocsc()
        Plow.sort_indices()
        if mode == "Cayley":
            fn = lambda x: x + 2.0 * sigma * lu.solve(P @ x)
        else:
            fn = lambda x: lu.solve(P @ x)
        self._op = Op.callback(n, fn)
        self._bop = Op.csc_sym(n, Plow.indptr, Plow.indices, Plow.data, lower=True)
        self.h = lib().oracle_symeigs_create_b(self._op.h, self._bop.h, nev, ncv, self.MODES[mode], float(sigma))
        if not self.h:
            raise ValueError(lib().oracle_last_error().decode())
        self.op, self.nev, self.ncv, self.n = self._op, nev, min(ncv, n), n


def hess_qr(H, shift):
    """UpperHessenbergQR (real): returns (Q, Q'HQ = RQ + sI)."""
    H = np.asfortranarray(H, dtype=np.float64)
    n = H.shape[0]
    Q, D = np.empty((n, n), order="F"), np.empty((n, n), order="F")
    _check(lib().oracle_hess_qr(n, _dp(H), shift, _dp(Q), _dp(D)))
    return Q, D


def double_shift_qr(H, s, t):
    """DoubleShiftQR: H^2 - sH + tI = QR; returns (Q, Q'HQ)."""
    H = np.asfortranarray(H, dtype=np.float64)
    n = H.shape[0]
    Q, D = np.empty((n, n), order="F"), np.empty((n, n), order="F")
    _check(lib().oracle_double_shift_qr(n, _dp(H), s, t, _dp(Q), _dp(D)))
    return Q, D


def hess_schur(H):
    H = np.asfortranarray(H, dtype=np.float64)
    n = H.shape[0]
    T, U = np.empty((n, n), order="F"), np.empty((n, n), order="F")
    _check(lib().oracle_hess_schur(n, _dp(H), _dp(T), _dp(U)))
    return T, U


def hess_eigen(H):
    H = np.asfortranarray(H, dtype=np.float64)
    n = H.shape[0]
    ev = np.empty(n, dtype=np.complex128)
    V = np.empty((n, n), dtype=np.complex128, order="F")
    _check(lib().oracle_hess_eigen(n, _dp(H), ev.ctypes.data_as(C.POINTER(C.c_double)), V.ctypes.data_as(C.POINTER(C.c_double))))
    return ev, V


class GenEigsSolver:
    """GenEigsSolver.h / GenEigsBase.h on the oracle (real matrices, complex results).  sigma: the op is already
    (A - sigma I)^{-1} and the Ritz values are mapped back as in GenEigsRealShiftSolver.h:52-58."""

    def __init__(self, op, nev, ncv, sigma=None, complex_shift=None):
        """complex_shift = (sigmar, sigmai, op_probe): GenEigsComplexShiftSolver.h — `op` is x -> Re((A - sigma I)^{-1} x) and
        op_probe the same operator at the real probe shift complex_shift_probe(sigmar)."""
        self.op, self.nev, self.ncv, self.n = op, nev, min(ncv, op.n), op.n
        self.h = lib().oracle_geneigs_create(op.h, nev, ncv)
        if not self.h:
            raise ValueError(lib().oracle_last_error().decode())
        if sigma is not None:
            lib().oracle_geneigs_set_shift_invert(self.h, float(sigma))
        if complex_shift is not None:
            sr, si, probe = complex_shift
            self._probe = probe  # keep the callback alive
            lib().oracle_geneigs_set_complex_shift(self.h, float(sr), float(si), probe.h)

    def init(self, v0=None):
        v0 = None if v0 is None else _f64(v0)
        _check(lib().oracle_geneigs_init(self.h, _dp(v0)))

    def compute(self, selection=LargestMagn, maxit=1000, tol=1e-10, sorting=LargestMagn):
        rc = lib().oracle_geneigs_compute(self.h, selection, maxit, tol, sorting)
        _check(rc)
        return rc

    def info(self):
        return lib().oracle_geneigs_info(self.h)

    def num_iterations(self):
        return lib().oracle_geneigs_num_iterations(self.h)

    def num_operations(self):
        return lib().oracle_geneigs_num_operations(self.h)

    def eigenvalues(self):
        out = np.empty(self.nev, dtype=np.complex128)
        cnt = lib().oracle_geneigs_eigenvalues(self.h, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out[:cnt].copy()

    def eigenvectors(self, nvec=None):
        nvec = self.nev if nvec is None else nvec
        out = np.zeros((self.n, max(nvec, 1)), dtype=np.complex128, order="F")
        cnt = lib().oracle_geneigs_eigenvectors(self.h, nvec, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out[:, :cnt].copy(order="F")

    def __del__(self):
        try:
            lib().oracle_geneigs_free(self.h)
        except Exception:
            pass


def time_restart_cycles(op, nev, ncv, selection=LargestMagn, tol=1e-10, cycles=1):
    """cpu_baseline helper: init() + factorize_from(1, ncv), then `cycles` restart cycles of the IRLM driver.
    Returns (seconds of the first sweep, its perform_op count, seconds of the cycles, their perform_op count, cycles run)."""
    out = np.zeros(5)
    _check(lib().oracle_symeigs_time_cycles(op.h, nev, ncv, selection, tol, cycles, _dp(out)))
    return float(out[0]), int(out[1]), float(out[2]), int(out[3]), int(out[4])


def time_lanczos_steps(op, ncv, nsteps):
    """cpu_baseline helper: wall seconds of init() + nsteps Lanczos steps, and the perform_op count."""
    nops = C.c_long()
    t = lib().oracle_symeigs_time_steps(op.h, ncv, nsteps, C.byref(nops))
    return t, nops.value
