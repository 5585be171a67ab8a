"""Workload for rocprofv3 --pmc passes: the int32 CSR product on the jittered band at n = 1e7, with x windows and with gathers
(tools/gpu_call.sh pmc <tag> tools/probes/pmc_jitter.py; tools/pmc_summarize.py calibrates FETCH_SIZE on the k_scale launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, torch
import spectra_amd as sa
from spectra_amd import workloads
n = int(os.environ.get("PROBE_N", 10_000_000))
ctx = sa.default_context()
A = workloads.jitter_band(n)
op = sa.SparseSymMatProd(sp.tril(A).tocsc(), ctx=ctx)
del A
x = torch.rand(n, dtype=torch.float64, device="cuda") - 0.5
y = torch.empty(n + 2, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
# calibration launches: k_scale moves exactly 8n bytes each way
band = sa.SparseSymMatProd.synth_band(n, ctx=ctx)
fac = sa.Factorization(band, 4, True)
fac.set_orth_mode("reference")
fac.init_random(0)
fac.factorize_from(1, 4)
del fac, band
for windows in (True, False):
    op.use_windows(windows)
    for _ in range(5):
        op.spmv_device(x.data_ptr(), y.data_ptr())
    ctx.sync()
print("done", op.spmv_format(), op.windows_info(), op.nnz())
