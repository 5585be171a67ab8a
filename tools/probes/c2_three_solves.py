"""Three default-mode C2 solves (no profiling events): the workload for a rocprofv3 kernel trace / tools/trace_gaps.py."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectra_amd as sa
ctx = sa.default_context()
op = sa.SparseSymMatProd.synth_band(int(os.environ.get("PROBE_N", 10_000_000)), ctx=ctx)
e = sa.SymEigsSolver(op, 20, 40)
for r in range(3):
    ctx.sync(); t0 = time.perf_counter()
    e.init(); nconv = e.compute(sa.SortRule.LargestMagn, 1000, 1e-11); e.eigenvectors(to_host=False); ctx.sync()
    print(json.dumps({"solve": r, "seconds": time.perf_counter() - t0, "nconv": int(nconv), "num_operations": int(e.num_operations())}), flush=True)
