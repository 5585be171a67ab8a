"""CPU restatement of the reference's block Davidson solver — TEST INFRASTRUCTURE ONLY (used by tests/; never by the
product).  numpy for the arithmetic; each step cites the reference file:line it follows.

Restated: JDSymEigsBase.h:28-187 (driver, sizes, compute / compute_with_guess), DavidsonSymEigsSolver.h:18-90 (diagonal,
unit-vector start, DPR correction), LinAlg/SearchSpace.h:24-93, LinAlg/RitzPairs.h:24-127, LinAlg/Orthogonalization.h:107-137
(twice_is_enough = 2 x [project out the old space, Householder QR of the new block]) and Util/SelectionRule.h argsort.
Third-party pieces of the reference: Eigen::SelfAdjointEigenSolver (here numpy.linalg.eigh, LAPACK) and
Eigen::HouseholderQR (here numpy.linalg.qr) — eigenvector / Q signs are implementation-defined in both, so parity is on
eigenvalues, residuals, convergence flags and iteration counts.
Pinned by: test/DavidsonSymEigs.cpp:69-123 (nconv == nev, Successful, ||AU - UD||_inf < 1e-10 on the reproducible sparse
fixture for LargestAlge / SmallestAlge) and numpy.linalg.eigvalsh of the same matrix (tests/test_oracle_davidson.py).
"""
import numpy as np

LargestMagn, LargestReal, LargestImag, LargestAlge, SmallestMagn, SmallestReal, SmallestImag, SmallestAlge, BothEnds = range(9)
Successful, NotComputed, NotConverging, NumericalIssue = range(4)


def argsort(selection, values):
    """Util/SelectionRule.h:225-287 for real values (std::sort is not stable; ties are not exercised by the tests)."""
    v = np.asarray(values, dtype=np.float64)
    if selection == LargestMagn:
        ind = np.argsort(-np.abs(v), kind="stable")
    elif selection in (LargestAlge, BothEnds):
        ind = np.argsort(-v, kind="stable")
    elif selection == SmallestMagn:
        ind = np.argsort(np.abs(v), kind="stable")
    elif selection == SmallestAlge:
        ind = np.argsort(v, kind="stable")
    else:
        raise ValueError("unsupported selection rule")
    if selection == BothEnds:  # :265-284
        n = len(ind)
        ind = np.array([ind[i // 2] if i % 2 == 0 else ind[n - 1 - i // 2] for i in range(n)], dtype=np.int64)
    return ind


def _jens_wehner(V, skip):
    """Orthogonalization.h:107-127: project the new block on the complement of the old one, then thin Householder Q."""
    if skip > 0:
        V[:, skip:] -= V[:, :skip] @ (V[:, :skip].T @ V[:, skip:])
    Q, _ = np.linalg.qr(V[:, skip:])
    V[:, skip:] = Q


class DavidsonSymEigsSolver:
    def __init__(self, A, nev, nvec_init=None, nvec_max=None):
        """A: anything with @ for a dense block and .diagonal() (numpy array or scipy sparse), symmetric."""
        self.A = A
        n = A.shape[0]
        self.n = n
        nvec_init = 2 * nev if nvec_init is None else nvec_init
        nvec_max = 10 * nev if nvec_max is None else nvec_max
        if nev < 1 or nev > n - 1:  # JDSymEigsBase.h:49-53
            raise ValueError("nev must satisfy 1 <= nev <= n - 1, n is the size of matrix")
        self.nev = nev
        self.max_size = nvec_max if nvec_max < n else 10 * nev  # :70-78
        self.init_size = nvec_init if nvec_init < n else 2 * nev
        self.corr_size = nev
        if n < self.max_size:  # initialize(), :55-66
            self.max_size = n
        if n < self.init_size + self.corr_size:
            self.init_size = n // 3
            self.corr_size = n // 3
        self.diag = np.asarray(A.diagonal(), dtype=np.float64).ravel()  # DavidsonSymEigsSolver.h:33-38
        self._info = NotComputed
        self.niter = 0

    def compute(self, selection=LargestMagn, maxit=100, tol=1e-10):
        order = argsort(selection, self.diag)  # DavidsonSymEigsSolver.h:47-58
        V0 = np.zeros((self.n, self.init_size))
        for k in range(self.init_size):
            V0[order[k], k] = 1.0
        return self.compute_with_guess(V0, selection, maxit, tol)

    def compute_with_guess(self, initial_space, selection=LargestMagn, maxit=100, tol=1e-10):
        V = np.array(initial_space, dtype=np.float64, order="F")  # SearchSpace.h:45-49
        AV = np.zeros((self.n, 0))
        self.nops = 0
        self.niter = 0
        vals = Ysm = X = R = None
        for self.niter in range(maxit):
            if V.shape[1] > self.max_size:  # JDSymEigsBase.h:148-153, SearchSpace.h:59-63
                AV = AV @ Ysm[:, :self.init_size]
                V = X[:, :self.init_size].copy()
            new = V.shape[1] - AV.shape[1]  # SearchSpace.h:51-57
            if new > 0:
                AV = np.hstack([AV, np.asarray(self.A @ V[:, -new:])])
                self.nops += new
            G = V.T @ AV  # RitzPairs.h:113-122
            try:
                vals, Ysm = np.linalg.eigh(0.5 * (G + G.T))
            except np.linalg.LinAlgError:
                self._info = NumericalIssue
                break
            X = V @ Ysm
            R = AV @ Ysm - X * vals
            ind = argsort(selection, vals)  # RitzPairs.h:41-52
            vals, Ysm, X, R = vals[ind], Ysm[:, ind], X[:, ind], R[:, ind]
            norms = np.linalg.norm(R, axis=0)  # :54-68
            self.root_converged = norms < tol
            if np.all(self.root_converged[:self.nev]):
                self._info = Successful
                break
            if self.niter == maxit - 1:
                self._info = NotConverging
                break
            corr = np.empty((self.n, self.corr_size))  # DavidsonSymEigsSolver.h:61-76
            with np.errstate(divide="ignore", invalid="ignore"):
                for k in range(self.corr_size):
                    corr[:, k] = R[:, k] / (vals[k] - self.diag)
            skip = V.shape[1]  # SearchSpace.h:65-70, Orthogonalization.h:130-137
            V = np.hstack([V, corr])
            _jens_wehner(V, skip)
            _jens_wehner(V, skip)
        self._vals, self._X = vals, X
        return int(np.sum(self.root_converged[:self.nev]))  # JDSymEigsBase.h:183

    def info(self):
        return self._info

    def num_iterations(self):
        return self.niter

    def num_operations(self):
        return self.nops

    def eigenvalues(self):
        return self._vals[:self.nev].copy()

    def eigenvectors(self):
        return self._X[:, :self.nev].copy()
