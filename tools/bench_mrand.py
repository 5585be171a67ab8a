"""Scattered-pattern SpMV (SURVEY.md 8d M-rand) at n = argv[1] (default 1e7): the tile kernel (format 3) and the int32 CSR kernel
on the same matrix, stand-alone.  One JSON line.  MISPEC_SPMV_TILES (auto | 1 | 0) is read from the environment."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch

import bench
import spectra_amd as sa

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
ctx = sa.default_context()
A = bench.m_rand_host(n)
tri = sp.tril(A).tocsc()
t0 = time.perf_counter()
op = sa.SparseSymMatProd(tri, ctx=ctx)
ingest = time.perf_counter() - t0
x = torch.rand(n, dtype=torch.float64, device="cuda") - 0.5
y3 = torch.empty(n + 2, dtype=torch.float64, device="cuda")
y0 = torch.empty(n + 2, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
out = {"n": n, "nnz": op.nnz(), "ingest_s": round(ingest, 3), "ingest_stages": {k: round(v, 3) for k, v in sa.last_ingest_info().items()}, "MISPEC_SPMV_TILES": os.environ.get("MISPEC_SPMV_TILES", "auto"), "tiles": op.tiles_info()}
for fmt, yy in ((3, y3), (0, y0)):
    op.set_spmv_format(fmt)
    op.spmv_time(x.data_ptr(), yy.data_ptr(), 3)
    ms = op.spmv_time(x.data_ptr(), yy.data_ptr(), 20)
    out[f"format{op.spmv_format()}_ms"] = round(ms, 4)
    out[f"format{op.spmv_format()}_frac_true_bytes"] = round(op.stored_bytes() / (ms * 1e-3) / 8e12, 4)
    out[f"format{op.spmv_format()}_frac_csr_bytes"] = round(op.algorithmic_bytes() / (ms * 1e-3) / 8e12, 4)
ctx.sync()
out["bit_identical"] = bool(torch.equal(y3[:n], y0[:n]))
print(json.dumps(out), flush=True)
