"""One reduction per lagged step against the two-reduction form, same box: C2 (n = 1e7) and small sizes (latency per operation)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import spectra_amd as sa
ctx = sa.default_context()
for n, reps in ((10_000_000, 3), (1_250_000, 5), (100_000, 10), (10_000, 10)):
    op = sa.SparseSymMatProd.synth_band(n, ctx=ctx) if n >= 300_000 else sa.SparseSymMatProd.synth_band(n, offsets=(1, 2, 3, 50, 51, 1500, 1501), ctx=ctx)
    for mode in ("onesweep-twored", "onesweep-onered", "onesweep-twored", "onesweep-onered"):
        e = sa.SymEigsSolver(op, 20, 40)
        e.set_orth_mode(mode)
        best = None
        for r in range(reps):
            ctx.sync(); t0 = time.perf_counter()
            e.init(); nconv = e.compute(sa.SortRule.LargestMagn, 1000, 1e-11); e.eigenvectors(to_host=False); ctx.sync()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        info = e.orth_info()
        print(json.dumps({"n": n, "mode": mode, "seconds": round(best, 5), "us_per_operation": round(1e6 * best / e.num_operations(), 2), "nconv": int(nconv),
                          "num_operations": int(e.num_operations()), "num_iterations": int(e.num_iterations()), "max_residual": float(e.residuals().max()),
                          "lagged_steps": info["lagged_steps"], "one_reduction_steps": info["one_reduction_steps"], "check_stops": info["check_stops"],
                          "state_stops": info["state_stops"], "lambda_max": float(e.eigenvalues().max())}), flush=True)
        del e
