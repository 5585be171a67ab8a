"""SpMV on a 3-D 7-point stencil (m^3 rows, default m = 215: 9.94 M rows, 69 M entries) in natural order, in random order
as it arrives (int32 CSR kernel / tiles), and in random order with the reordering at ingest (reverse Cuthill-McKee) —
stand-alone and inside a SymEigsSolver loop.  One JSON line per variant; fractions are of 8 TB/s on the
bytes the kernel has to move (`stored_bytes`).

    python tools/bench_stencil.py [m]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch

import spectra_amd as sa

m = int(sys.argv[1]) if len(sys.argv) > 1 else 215
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
ctx = sa.default_context()


def stencil7(m, seed=0):
    I = sp.identity(m, format="csr")
    T = sp.diags([np.ones(m - 1), np.ones(m - 1)], [-1, 1], format="csr")
    A = (sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T) + 6 * sp.identity(m ** 3)).tocsr()
    rng = np.random.default_rng(seed)
    A.data[:] = rng.uniform(-0.5, 0.5, A.nnz)
    A = (sp.tril(A) + sp.tril(A, -1).T).tocsr()
    A.sort_indices()
    return A


def measure(name, op, extra):
    n = op.rows()
    x = torch.rand(n, dtype=torch.float64, device="cuda") - 0.5
    y = torch.empty(n + 2, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    op.spmv_time(x.data_ptr(), y.data_ptr(), 3)
    alone = op.spmv_time(x.data_ptr(), y.data_ptr(), 20)
    e = sa.SymEigsSolver(op, 10, 30)
    e.profile(2)
    e.init()
    t0 = time.perf_counter()
    nconv = e.compute(sa.SortRule.LargestAlge, 8, 1e-10)   # a bounded number of restarts: the in-loop time needs no convergence
    ctx.sync()
    dt = time.perf_counter() - t0
    p = e.get_profile()
    inloop = p["ms_spmv"] / max(p["n_spmv"], 1)
    stored = op.stored_bytes()
    out = {"variant": name, "n": n, "nnz": op.nnz(), "spmv_format": op.spmv_format(), "reordering": op.reordering_info(),
           "standalone_ms": round(alone, 4), "standalone_frac": round(stored / (alone * 1e-3) / 8e12, 4),
           "in_loop_ms": round(inloop, 4), "in_loop_frac": round((stored + 16.0 * n) / (inloop * 1e-3) / 8e12, 4),
           "in_loop_launches": int(p["n_spmv"]), "solve_8_restarts_s": round(dt, 3), "nconv": int(nconv)}
    out.update(extra)
    print(json.dumps(out), flush=True)


A = stencil7(m)
n = A.shape[0]
variants = []
if not only or "natural" in only:
    t0 = time.perf_counter()
    op = sa.SparseSymMatProd(sp.tril(A).tocsc(), ctx=ctx)
    measure("natural order", op, {"ingest_s": round(time.perf_counter() - t0, 2)})
    del op
perm = np.random.default_rng(1).permutation(n)
B = A[perm][:, perm].tocsr()
B.sort_indices()
Bl = sp.tril(B).tocsc()
del A
for name, kw, env in (("random order, reordered at ingest (RCM)", {}, {}),
                      ("random order, as it comes: tiles", {"reorder": "none"}, {}),
                      ("random order, as it comes: int32 CSR kernel", {"reorder": "none"}, {"MISPEC_SPMV_TILES": "0", "MISPEC_SPMV_STAGED": "0"})):
    key = name.split(",")[1].split(":")[-1].strip().split(" ")[0]
    if only and not any(k in name for k in only):
        continue
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    t0 = time.perf_counter()
    op = sa.SparseSymMatProd(Bl, ctx=ctx, **kw)
    ingest = time.perf_counter() - t0
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    measure(name, op, {"ingest_s": round(ingest, 2)})
    del op
