"""Workload for rocprofv3 --pmc passes on the staged SpMV (format 4) and the tiles (format 3) on M-rand (n = PROBE_N,
default 1e7): five stand-alone launches of each, then a Lanczos factorisation in the reference flow with the staged format (its
k_scale_step launches calibrate FETCH_SIZE, the product runs with the fused epilogue).

    MISPEC_SPMV_TILES=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -o fetch -- python tools/pmc_probe_staged.py
    MISPEC_SPMV_TILES=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT -o write -- python tools/pmc_probe_staged.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MISPEC_SPMV_TILES", "1")
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
for fmt in (4, 3):
    op.set_spmv_format(fmt)
    for _ in range(5):
        op.spmv_device(x.data_ptr(), y.data_ptr())
    ctx.sync()
    print("format", op.spmv_format(), "ms", op.spmv_time(x.data_ptr(), y.data_ptr(), 5))
op.set_spmv_format(4)
fac = sa.Factorization(op, 12, True)
fac.set_orth_mode("reference")
fac.init_random(0)
fac.factorize_from(1, 12)
ctx.sync()
print("probe done: format", op.spmv_format(), "nops =", fac.num_operations())
