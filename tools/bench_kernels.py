"""Kernel-level timing at BASELINE.json's size: one full Lanczos factorisation (39 steps) + one restart, per-family
HIP-event times from the library's own profile.  Used to compare tuning variants (env knobs) on the GPU box.

    python tools/bench_kernels.py [n] [reps]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import spectra_amd as sa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
m = 40
ctx = sa.default_context()
op = sa.SparseSymMatProd.synth_band(n, ctx=ctx)
best = None
for r in range(reps + 1):
    fac = sa.Factorization(op, m, True)
    fac.profile(True)
    fac.init_random(0)
    fac.factorize_from(1, m)
    ev, U = fac.tridiag_eigen()
    order = np.argsort(-np.abs(ev))
    fac.restart_sym(ev[order][25:])
    p = fac.get_profile()
    if r == 0:
        continue  # warm-up
    if best is None or p["ms_vtf"] + p["ms_gemv"] < best["ms_vtf"] + best["ms_gemv"]:
        best = p
vec = 8.0 * n
passes_resid = sum(i1 + 2 for i1 in range(2, m + 1)) + 3  # RESID_VTF launches (+ init)
gb = lambda passes, ms: passes * vec / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
print({k: round(v, 3) if isinstance(v, float) else v for k, v in best.items()})
print(f"spmv   : {best['ms_spmv'] / best['n_spmv']:.4f} ms/launch = {best['spmv_bytes'] / (best['ms_spmv'] / best['n_spmv'] * 1e-3) / 1e9:.0f} GB/s")
print(f"resid  : {best['ms_vtf']:.2f} ms total over {best['n_vtf']} launches ~ {gb(passes_resid, best['ms_vtf']):.0f} GB/s (incl. reduce kernels)")
print(f"correct: {best['ms_gemv']:.2f} ms total over {best['n_gemv']} launches")
print(f"compress (V*Q 40->26 cols + axpby): {best['ms_compress']:.3f} ms ~ {gb(40 + 26 + 3, best['ms_compress']):.0f} GB/s")
print(f"small  : {best['ms_small']:.3f} ms over {best['n_small']} launches")
