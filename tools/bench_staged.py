"""M-rand at the headline size: the staged format (4) against the tiles (3) and the int32 CSR kernel (0), stand-alone and inside
a solver loop (restarts capped), one JSON line each.      MISPEC_SPMV_STAGED=1 python tools/bench_staged.py [n]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MISPEC_SPMV_STAGED", "1")
import numpy as np

import bench
import spectra_amd as sa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = sa.default_context()
t0 = time.time()
A = bench.m_rand_host(n)
t_gen = time.time() - t0
t0 = time.time()
import scipy.sparse as sp

op = sa.SparseSymMatProd(sp.tril(A).tocsc(), ctx=ctx)
t_ingest = time.time() - t0
print(json.dumps({"n": n, "nnz": int(A.nnz), "host_generation_s": t_gen, "ingest_s": t_ingest, "ingest_stages": sa.last_ingest_info(),
                  "staged": op.staged_info(), "tiles": op.tiles_info()}), flush=True)
csr_bytes = 12.0 * A.nnz + 20.0 * n + 4
for fmt in [int(a) for a in os.environ.get("BENCH_FORMATS", "4,3,0").split(",")]:
    op.set_spmv_format(fmt)
    if op.spmv_format() != fmt:
        continue
    ms = bench.standalone_ms(op, n, 30)
    e = sa.SymEigsSolver(op, 20, 40)
    e.profile(2)
    e.init()
    p0 = e.get_profile()
    t0 = time.perf_counter()
    nconv = e.compute(sa.SortRule.LargestMagn, 12, 1e-11)
    ctx.sync()
    dt = time.perf_counter() - t0
    p1 = e.get_profile()
    nsp = p1["n_spmv"] - p0["n_spmv"]
    in_loop = (p1["ms_spmv"] - p0["ms_spmv"]) / max(1, nsp)
    print(json.dumps({"format": fmt, "stored_bytes": op.stored_bytes(), "standalone_ms": ms, "in_loop_ms": in_loop,
                      "in_loop_frac_of_8TBs_on_csr_bytes": csr_bytes / (in_loop * 1e-3) / 8e12 if in_loop else None,
                      "solve_12_restarts_s": dt, "nconv": int(nconv), "num_operations": int(e.num_operations())}), flush=True)
    del e
op.set_spmv_format(-1)
