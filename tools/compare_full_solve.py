"""A complete solve on both sides, no extrapolation (SURVEY.md §8d "CPU baseline", item ii): the GPU library and the
CPU oracle run SymEigsSolver on the same M-band matrix (same generator, same SimpleRandom(0) start vector, same
selection / tol) and the results are compared pair by pair.

    python tools/compare_full_solve.py [n] [nev] [ncv] [tol]

Defaults: n = 1e6, nev = 20, ncv = 40, tol = 1e-11 (BASELINE.json's k / ncv at a size one host core finishes in
well under a minute).  Prints one JSON object: wall times, operation / restart counts of both sides, max |d lambda|,
max residual of each side.  The oracle is test infrastructure: this script is a measurement tool, not product code.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

import oracle as O
import spectra_amd as sa

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
nev = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ncv = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-11

ctx = sa.default_context()
op = sa.SparseSymMatProd.synth_band(n, ctx=ctx)
best = None
for rep in range(3):
    eigs = sa.SymEigsSolver(op, nev, ncv)
    ctx.sync()
    t0 = time.perf_counter()
    eigs.init()
    nconv = eigs.compute(sa.SortRule.LargestMagn, 1000, tol)
    X = eigs.eigenvectors(to_host=False)
    ctx.sync()
    dt = time.perf_counter() - t0
    if rep > 0 and (best is None or dt < best):
        best = dt
ev_gpu = eigs.eigenvalues()
res_gpu = eigs.residuals()

rp, ci, v = O.synth_band_csr(n)
A = sp.csr_matrix((v, ci, rp), shape=(n, n))
oop = O.Op.csr(n, n, rp, ci, v)
oe = O.SymEigsSolver(oop, nev, ncv)
t0 = time.perf_counter()
oe.init()
nconv_cpu = oe.compute(O.LargestMagn, 1000, tol)
Xc = oe.eigenvectors()
t_cpu = time.perf_counter() - t0
ev_cpu = oe.eigenvalues()
res_cpu = np.linalg.norm(A @ Xc - Xc * ev_cpu, axis=0) / np.linalg.norm(Xc, axis=0)

print(json.dumps({
    "n": n, "nev": nev, "ncv": ncv, "tol": tol, "selection": "LargestMagn",
    "gpu": {"seconds": best, "nconv": int(nconv), "num_operations": int(eigs.num_operations()),
            "num_iterations": int(eigs.num_iterations()), "max_residual": float(res_gpu.max())},
    "cpu_oracle_1_thread": {"seconds": t_cpu, "nconv": int(nconv_cpu), "num_operations": int(oe.num_operations()),
                            "num_iterations": int(oe.num_iterations()), "max_residual": float(res_cpu.max())},
    "max_abs_dlambda": float(np.abs(ev_gpu - ev_cpu).max()),
    "speedup_reported_only": t_cpu / best,
}))
