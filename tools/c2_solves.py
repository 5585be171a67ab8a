"""C2-only solves for a clean rocprofv3 trace of ONE SpMV kernel: BASELINE.json configs[1] (n = 1e7 M-band,
nev 20, ncv 40, LargestMagn, tol 1e-11), the default one-sweep flow, nothing else in the process.

    python tools/c2_solves.py [--format F] [--solves S] [--orth MODE]     F: -1 automatic (diagonal storage), 0 int32 CSR, 1 offset codes

Prints one JSON line: format, solves, seconds per solve, operations, the SpMV's HIP-event time per launch (profile level 2, one
more solve) and both byte counts, so that tests/test_profiles_roofline.py can recompute the fractions from the committed files.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spectra_amd as sa

p = argparse.ArgumentParser()
p.add_argument("--size", type=int, default=10_000_000)
p.add_argument("--format", type=int, default=-1)
p.add_argument("--solves", type=int, default=3)
p.add_argument("--orth", default=None)
p.add_argument("--nev", type=int, default=20)
p.add_argument("--ncv", type=int, default=40)
a = p.parse_args()
ctx = sa.default_context()
op = sa.SparseSymMatProd.synth_band(a.size, ctx=ctx)
if a.format >= 0:
    op.set_spmv_format(a.format)
e = sa.SymEigsSolver(op, a.nev, a.ncv)
if a.orth:
    e.set_orth_mode(a.orth)


def solve():
    e.init()
    nconv = e.compute(sa.SortRule.LargestMagn, 1000, 1e-11)
    e.eigenvectors(to_host=False)
    return nconv


solve()
ctx.sync()
t0 = time.perf_counter()
for _ in range(a.solves):
    nconv = solve()
ctx.sync()
s_per_solve = (time.perf_counter() - t0) / a.solves
e.profile(2)
p0 = e.get_profile()
solve()
p1 = e.get_profile()
e.profile(0)
n_spmv = p1["n_spmv"] - p0["n_spmv"]
ms_spmv = (p1["ms_spmv"] - p0["ms_spmv"]) / max(n_spmv, 1)
nnz = int(op.nnz())
n = a.size
print(json.dumps({"n": n, "nnz": nnz, "format": int(op.spmv_format()), "solves_traced": a.solves + 2, "seconds_per_solve": s_per_solve,
                  "eigenpairs_per_s": a.nev / s_per_solve, "nconv": int(nconv), "num_operations": int(e.num_operations()),
                  "num_iterations": int(e.num_iterations()), "spmv_ms_per_launch_hip_events": ms_spmv, "spmv_launches_timed": int(n_spmv),
                  "csr_bytes_survey_8d": 12 * nnz + 20 * n + 4, "max_residual": float(e.residuals().max())}))
