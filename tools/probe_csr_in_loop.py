"""In-loop figures of the CSR SpMV kernels on the headline matrix: one-sweep solves with the format forced (0 int32, 1 offset codes).
    python tools/probe_csr_in_loop.py [format ...]      prints one JSON line per format"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spectra_amd as sa

ctx = sa.default_context()
op = sa.SparseSymMatProd.synth_band(10_000_000, ctx=ctx)
for fmt in [int(a) for a in sys.argv[1:]] or [0, 1]:
    op.set_spmv_format(fmt)
    e = sa.SymEigsSolver(op, 20, 40)
    e.set_orth_mode("onesweep")
    e.profile(2)
    for rep in range(2):
        p0 = e.get_profile()
        ctx.sync()
        t0 = time.perf_counter()
        e.init()
        nconv = e.compute(sa.SortRule.LargestMagn, 1000, 1e-11)
        e.eigenvectors(to_host=False)
        ctx.sync()
        dt = time.perf_counter() - t0
    p1 = e.get_profile()
    n = p1["n_spmv"] - p0["n_spmv"]
    print(json.dumps({"format": op.spmv_format(), "spmv_ms_in_loop": (p1["ms_spmv"] - p0["ms_spmv"]) / n, "seconds_per_solve": dt,
                      "eigenpairs_per_s": nconv / dt, "num_operations": e.num_operations()}), flush=True)
    del e
op.set_spmv_format(-1)
