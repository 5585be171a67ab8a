"""Workload for rocprofv3 --pmc passes on the scattered-pattern SpMV (SURVEY.md 8d M-rand, n = PROBE_N, default 1e7): five
stand-alone launches of the tile kernel (format 3) and five of the int32 CSR kernel on the same matrix.

    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d OUT -o l2 -- python tools/pmc_probe_mrand.py
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -o fetch -- python tools/pmc_probe_mrand.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scipy.sparse as sp
import torch

import bench
import spectra_amd as sa

n = int(os.environ.get("PROBE_N", 10_000_000))
ctx = sa.default_context()
A = bench.m_rand_host(n)
op = sa.SparseSymMatProd(sp.tril(A).tocsc(), ctx=ctx)
x = torch.rand(n, dtype=torch.float64, device="cuda") - 0.5
y = torch.empty(n + 2, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for fmt in (3, 0):
    op.set_spmv_format(fmt)
    for _ in range(5):
        op.spmv_device(x.data_ptr(), y.data_ptr())
    ctx.sync()
    print("format", op.spmv_format(), "ms", op.spmv_time(x.data_ptr(), y.data_ptr(), 5))
