"""Where does the diagonal-storage product differ from the int32 CSR product on the natural-order 7-point stencil?"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, torch
import spectra_amd as sa
from spectra_amd import workloads
m = int(sys.argv[1]) if len(sys.argv) > 1 else 215
A = workloads.stencil7(m)
n = A.shape[0]
ctx = sa.default_context()
op = sa.SparseSymMatProd(sp.tril(A).tocsc(), ctx=ctx)
x = torch.rand(n, dtype=torch.float64, device="cuda") - 0.5
ys = {}
for fmt in (2, 1, 0):
    op.set_spmv_format(fmt)
    y = torch.zeros(n + 2, dtype=torch.float64, device="cuda")
    op.spmv_device(x.data_ptr(), y.data_ptr()); ctx.sync()
    ys[op.spmv_format()] = y[:n].cpu().numpy()
xr = x.cpu().numpy()
ref = np.zeros(n)
# CSR row-dot in storage order, one accumulator, product rounded first
ip, ix, iv = A.indptr, A.indices, A.data
prod = iv * xr[ix]
maxlen = int(np.diff(ip).max())
for k in range(maxlen):
    idx = ip[:-1] + k
    ok = idx < ip[1:]
    ref[ok] = ref[ok] + prod[idx[ok]]
out = {"formats": list(ys)}
for f, y in ys.items():
    d = np.nonzero(y != ref)[0]
    out["fmt%d_mismatch_rows" % f] = int(d.size)
    if d.size:
        out["fmt%d_first_rows" % f] = d[:8].tolist()
        out["fmt%d_first_vals" % f] = [[float(y[i]), float(ref[i])] for i in d[:4]]
        out["fmt%d_rowlen" % f] = [int(ip[i + 1] - ip[i]) for i in d[:8]]
        out["fmt%d_cols_of_first" % f] = (ix[ip[d[0]]:ip[d[0] + 1]] - d[0]).tolist()
print(json.dumps(out))
