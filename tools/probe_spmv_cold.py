"""Does the SpMV's in-loop penalty come from the memory system's state?  Stand-alone launches of each storage format of the
headline matrix (a) back to back, (b) each one after a pass over 3.2 GB of other data (what a Lanczos step's basis pass leaves
behind: caches, MALL and TLBs hold the basis, not the matrix).  One JSON line per format."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import spectra_amd as sa

n = 10_000_000
ctx = sa.default_context()
op = sa.SparseSymMatProd.synth_band(n, ctx=ctx)
x = torch.rand(n, dtype=torch.float64, device="cuda") - 0.5
y = torch.empty(n + 2, dtype=torch.float64, device="cuda")
V = torch.rand(40 * n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for fmt in (0, 1, 2):
    op.set_spmv_format(fmt)
    op.spmv_time(x.data_ptr(), y.data_ptr(), 5)
    warm = op.spmv_time(x.data_ptr(), y.data_ptr(), 30)
    cold = []
    for rep in range(12):
        s = float(V.sum())  # 3.2 GB streamed (and a host sync)
        cold.append(op.spmv_time(x.data_ptr(), y.data_ptr(), 1))
    cold2 = []
    for rep in range(12):
        V.mul_(1.0000001)   # 3.2 GB read AND written
        torch.cuda.synchronize()
        cold2.append(op.spmv_time(x.data_ptr(), y.data_ptr(), 1))
    print(json.dumps({"format": op.spmv_format(), "back_to_back_ms": warm, "after_3.2GB_read_ms": sorted(cold)[len(cold) // 2],
                      "after_3.2GB_read_write_ms": sorted(cold2)[len(cold2) // 2], "single_launch_samples": [round(c, 4) for c in cold[:6]]}), flush=True)
op.set_spmv_format(-1)
