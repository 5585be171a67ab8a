"""Same-process, interleaved A/B of the orthogonalisation modes on the headline workload (BASELINE.json configs[1]): one matrix,
one solver object per mode, the modes timed in turn `--reps` times so that clock / box drift hits all of them alike.

    python tools/ab_orth.py [--size N] [--reps R] [--solves S] mode [mode ...]      modes: spectra_amd.ORTH_MODES keys
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
p.add_argument("--reps", type=int, default=4)
p.add_argument("--solves", type=int, default=2)
p.add_argument("--nev", type=int, default=20)
p.add_argument("--ncv", type=int, default=40)
p.add_argument("modes", nargs="+")
a = p.parse_args()
ctx = sa.default_context()
op = sa.SparseSymMatProd.synth_band(a.size, ctx=ctx)
solvers = {}
for mode in a.modes:
    e = sa.SymEigsSolver(op, a.nev, a.ncv)
    e.set_orth_mode(mode)
    solvers[mode] = e


def solve(e):
    e.init()
    nconv = e.compute(sa.SortRule.LargestMagn, 1000, 1e-11)
    e.eigenvectors(to_host=False)
    return nconv


for e in solvers.values():
    solve(e)  # warm-up
times = {m: [] for m in a.modes}
for rep in range(a.reps):
    for mode, e in solvers.items():
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(a.solves):
            nconv = solve(e)
        ctx.sync()
        times[mode].append((time.perf_counter() - t0) / a.solves)
for mode, e in solvers.items():
    ts = sorted(times[mode])
    print(json.dumps({"mode": mode, "ms_per_solve_min": round(1e3 * ts[0], 1), "ms_per_solve_median": round(1e3 * ts[len(ts) // 2], 1),
                      "ms_per_solve_all": [round(1e3 * t, 1) for t in times[mode]], "eigenpairs_per_s_best": round(a.nev / ts[0], 3),
                      "num_operations": e.num_operations(), "max_residual": float(e.residuals().max()),
                      "orth_info": e.orth_info()}), flush=True)
