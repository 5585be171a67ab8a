"""In-loop time of the int32 CSR SpMV on the headline matrix, by orthogonalisation flow (the library's MISPEC_CSR_CHUNK experiment
knob picks the chunk size per PROCESS, so this script is run once per chunk setting).
    MISPEC_CSR_CHUNK=2|4 python tools/probe_csr_variants.py [format ...]     one JSON line per (format, flow)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spectra_amd as sa

ctx = sa.default_context()
op = sa.SparseSymMatProd.synth_band(10_000_000, ctx=ctx)
for fmt in [int(a) for a in sys.argv[1:]] or [0]:
    op.set_spmv_format(fmt)
    for mode in ("reference", "onesweep", "reference", "onesweep"):
        e = sa.SymEigsSolver(op, 20, 40)
        e.set_orth_mode(mode)
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
        print(json.dumps({"chunk_knob": os.environ.get("MISPEC_CSR_CHUNK", "auto"), "format": op.spmv_format(), "mode": mode,
                          "spmv_ms_in_loop": (p1["ms_spmv"] - p0["ms_spmv"]) / n, "seconds_per_solve": dt,
                          "num_operations": e.num_operations()}), flush=True)
        del e
op.set_spmv_format(-1)
