"""Stand-alone SpMV bandwidth on the two synthetic patterns of SURVEY.md §8(d):

  M-band  the benchmark matrix (15 nnz/row at fixed offsets; x gathers stay inside a sliding window)
  M-rand  every row i has 7 partners at uniformly random columns, symmetrised (row degrees vary around 15):
          the x gathers have no locality, which is the part of the 8 TB/s roofline a banded matrix does not test

    python tools/bench_spmv_patterns.py [n] [reps]

Prints one JSON object per pattern: ms per launch, algorithmic GB/s (12 nnz + 4(rows+1) + 8 cols + 8 rows) and the
fraction of 8 TB/s.  The random pattern is generated on the host with numpy/scipy (about a minute at n = 1e7).
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ctx = sa.default_context()


def measure(name, op, extra=None):
    x = torch.rand(n, dtype=torch.float64, device="cuda") - 0.5
    y = torch.empty(n + 2, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    op.spmv_time(x.data_ptr(), y.data_ptr(), 5)
    ms = op.spmv_time(x.data_ptr(), y.data_ptr(), reps)
    gbps = op.algorithmic_bytes() / (ms * 1e-3) / 1e9
    out = {"pattern": name, "n": n, "nnz": op.nnz(), "ms_per_launch": ms, "gbps": gbps, "frac_of_8TBps": gbps / 8000.0}
    if extra:
        out.update(extra)
    print(json.dumps(out), flush=True)
    return y


measure("M-band", sa.SparseSymMatProd.synth_band(n, ctx=ctx))

t0 = time.perf_counter()
rng = np.random.default_rng(20240607)
rows = np.repeat(np.arange(n, dtype=np.int64), 7)
cols = rng.integers(0, n, size=rows.size, dtype=np.int64)
vals = rng.uniform(-0.5, 0.5, size=rows.size)
U = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
U.sum_duplicates()
A = (U + U.T + sp.diags(rng.uniform(-0.5, 0.5, n))).tocsr()
A.sort_indices()
gen_s = time.perf_counter() - t0
op = sa.SparseGenMatProd(A, ctx=ctx)
y = measure("M-rand", op, {"host_generation_seconds": gen_s})
# spot check against scipy on the same x is not possible (x was random on the device); check a fixed x instead
x = np.linspace(-1.0, 1.0, n)
err = np.abs(op.perform_op(x) - A @ x).max()
print(json.dumps({"pattern": "M-rand", "max_abs_err_vs_scipy": float(err)}))
