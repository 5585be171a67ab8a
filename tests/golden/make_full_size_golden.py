"""Generates tests/golden/full_size_{c2,c4,c5}.json — the CPU oracle's COMPLETE solves of BASELINE.json's
configs[1], [3] and [4] at their full sizes (test infrastructure; the oracle is the Eigen-free restatement of the
reference in oracle/, one host thread like the reference).

    python tests/golden/make_full_size_golden.py c2     # n = 1e7, k = 20, ncv = 40          (about 5-6 min, 6 GB)
    python tests/golden/make_full_size_golden.py c4     # n = 5e6 non-symmetric, k = 10, ncv = 30
    python tests/golden/make_full_size_golden.py c5     # n = 2e6 banded, shift-and-invert sigma = 0, k = 6, ncv = 20 (scipy splu
                                                        # stands in for Eigen::SparseLU behind the oracle's callback operator)

Each file records: eigenvalues (repr-exact doubles; complex as [re, im]), nconv, num_operations, num_iterations, the
residual of every pair computed with scipy's SpMV (an implementation independent of both the oracle's and the
product's), a probe of the matrix (y = A x for the SimpleRandom(0) vector: the sum, the sum of |y| and 8 spot entries, so the
GPU test can also assert it is looking at the same matrix), and the wall time of the solve on this host.
The `-m gpu` tests compare the HIP solve with these files (tests/test_gpu_fullsize.py).
"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as O  # noqa: E402

TOL = 1e-11
SPOTS = [0, 1, 2, 999, 1000, 100001, -2, -1]


def probe(A, n):
    x = O.simple_random(n, 0)
    y = A @ x
    return {"sum": float(np.sum(y)), "abs_sum": float(np.sum(np.abs(y))), "spots": [float(y[i]) for i in SPOTS]}


def banded_spd(n, b, seed):  # the matrix of tests/test_gpu_shift.py::test_config5_banded
    rng = np.random.default_rng(seed)
    diags = [rng.uniform(-0.5, 0.5, n - d) for d in range(1, b + 1)]
    return sp.diags([rng.uniform(-0.5, 0.5, n) + b + 0.5] + diags + diags, [0] + list(range(1, b + 1)) + [-d for d in range(1, b + 1)],
                    format="csc")


def c2(n=10_000_000, nev=20, ncv=40):
    rp, ci, v = O.synth_band_csr(n)
    A = sp.csr_matrix((v, ci, rp), shape=(n, n))
    s = O.SymEigsSolver(O.Op.csr(n, n, rp, ci, v), nev, ncv)
    t0 = time.perf_counter()
    s.init()
    nconv = s.compute(O.LargestMagn, 1000, TOL)
    X = s.eigenvectors()
    dt = time.perf_counter() - t0
    ev = s.eigenvalues()
    res = np.linalg.norm(A @ X - X * ev, axis=0) / np.linalg.norm(X, axis=0)
    return {"config": "BASELINE.json configs[1]: SymEigsSolver, M-band symmetric", "n": n, "nev": nev, "ncv": ncv, "selection": "LargestMagn",
            "tol": TOL, "nconv": int(nconv), "info": int(s.info()), "num_operations": int(s.num_operations()),
            "num_iterations": int(s.num_iterations()), "eigenvalues": [float(x) for x in ev], "residuals_scipy": [float(r) for r in res],
            "matrix_probe": probe(A, n), "nnz": int(len(v)), "oracle_seconds_one_thread": dt}


def c4(n=5_000_000, nev=10, ncv=30):
    rp, ci, v = O.synth_band_csr(n, symmetric=False)
    A = sp.csr_matrix((v, ci, rp), shape=(n, n))
    s = O.GenEigsSolver(O.Op.csr(n, n, rp, ci, v), nev, ncv)
    t0 = time.perf_counter()
    s.init()
    nconv = s.compute(O.LargestMagn, 1000, TOL)
    X = s.eigenvectors()
    dt = time.perf_counter() - t0
    ev = s.eigenvalues()
    res = np.linalg.norm(A @ X - X * ev, axis=0) / np.linalg.norm(X, axis=0)
    return {"config": "BASELINE.json configs[3]: GenEigsSolver, M-band non-symmetric", "n": n, "nev": nev, "ncv": ncv,
            "selection": "LargestMagn", "tol": TOL, "nconv": int(nconv), "info": int(s.info()),
            "num_operations": int(s.num_operations()), "num_iterations": int(s.num_iterations()),
            "eigenvalues": [[float(x.real), float(x.imag)] for x in ev], "residuals_scipy": [float(r) for r in res],
            "matrix_probe": probe(A, n), "nnz": int(len(v)), "oracle_seconds_one_thread": dt}


def c5(n=2_000_000, nev=6, ncv=20, b=3, seed=5):
    A = banded_spd(n, b, seed)
    lu = spla.splu(A.tocsc())
    s = O.SymEigsSolver(O.Op.callback(n, lu.solve), nev, ncv, sigma=0.0)
    t0 = time.perf_counter()
    s.init()
    nconv = s.compute(O.LargestMagn, 1000, TOL)
    X = s.eigenvectors()
    dt = time.perf_counter() - t0
    ev = s.eigenvalues()
    res = np.linalg.norm(A @ X - X * ev, axis=0) / np.linalg.norm(X, axis=0)
    return {"config": "BASELINE.json configs[4]: SymEigsShiftSolver sigma = 0, banded half-bandwidth 3 (numpy default_rng(5))", "n": n,
            "nev": nev, "ncv": ncv, "selection": "LargestMagn", "tol": TOL, "sigma": 0.0, "nconv": int(nconv), "info": int(s.info()),
            "num_operations": int(s.num_operations()), "num_iterations": int(s.num_iterations()),
            "eigenvalues": [float(x) for x in ev], "residuals_scipy": [float(r) for r in res], "matrix_probe": probe(A.tocsr(), n),
            "nnz": int(A.nnz), "oracle_seconds_one_thread": dt}


if __name__ == "__main__":
    for name in sys.argv[1:] or ["c2", "c4", "c5"]:
        rec = {"c2": c2, "c4": c4, "c5": c5}[name]()
        rec["generator"] = "tests/golden/make_full_size_golden.py " + name
        with open(os.path.join(HERE, f"full_size_{name}.json"), "w") as f:
            json.dump(rec, f, indent=1)
        print(name, rec["nconv"], rec["num_operations"], rec["num_iterations"], f"{rec['oracle_seconds_one_thread']:.1f}s",
              max(rec["residuals_scipy"]))
