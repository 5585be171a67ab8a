"""Validate the x-window table of a matrix on the host and locate product mismatches of k_spmv_csr_win."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
import spectra_amd as sa
from spectra_amd import workloads
import oracle as O

m = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = workloads.stencil7(m)
perm = np.random.default_rng(1).permutation(B.shape[0])
A = B[perm][:, perm].tocsr(); A.sort_indices()
ctx = sa.default_context()
op = sa.SparseSymMatProd(sp.tril(A).tocsc(), ctx=ctx)
print(json.dumps({"n": A.shape[0], "format": op.spmv_format(), "reordering": op.reordering(), "windows": op.windows_info()}))
rp, ci, v = op.to_host_csr() if op.reordering() == "none" else (None, None, None)
# the stored (permuted) matrix
p = op.permutation()
S = A[p][:, p].tocsr(); S.sort_indices()
T = op.windows_table()
n = S.shape[0]
bad_blocks = []
for b in range(T.shape[0]):
    rec = T[b]
    nw, far, total = rec[0] & 255, rec[0] >> 8, rec[1]
    st, ad, en = rec[4:12].astype(np.int64), rec[12:20].astype(np.int64), rec[20:28].astype(np.int64)
    cols = S.indices[S.indptr[b * 256]: S.indptr[min(n, (b + 1) * 256)]].astype(np.int64)
    inside = np.zeros(cols.size, bool)
    base = 0
    ok = True
    for w in range(nw):
        ok &= (st[w] % 16 == 0) and (st[w] + ad[w] == base) and (en[w] > st[w]) and (w == 0 or st[w] >= en[w - 1])
        base += en[w] - st[w]
        inside |= (cols >= st[w]) & (cols < en[w])
    ok &= base == total
    ok &= bool(np.all(st[nw:] == 0x3fffffff))
    ok &= (int(inside.sum()) == rec[2])
    if not far:
        ok &= bool(inside.all())
    if not ok:
        bad_blocks.append(b)
print(json.dumps({"blocks": int(T.shape[0]), "invalid_records": bad_blocks[:10], "far_blocks": int((T[:, 0] >> 8).sum()), "max_windows": int((T[:, 0] & 255).max())}))
x = O.simple_random(n, 0)
ref = O.Op.csr(n, n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data).perform_op(x)
for windows in (True, False):
    op.use_windows(windows)
    y = op.perform_op(x)
    d = np.nonzero(y != ref)[0]
    out = {"windows": windows, "mismatch_rows": int(d.size)}
    if d.size:
        inv = np.empty(n, np.int64); inv[p] = np.arange(n)
        stored_rows = np.sort(inv[d])
        blocks = np.unique(stored_rows // 256)
        out["stored_rows"] = stored_rows[:10].tolist()
        out["blocks"] = blocks[:10].tolist()
        out["n_blocks"] = int(blocks.size)
        b = int(blocks[0])
        out["record"] = T[b].tolist()
        r = int(stored_rows[0])
        out["row_cols"] = S.indices[S.indptr[r]:S.indptr[r + 1]].tolist()
        out["block_entry_range"] = [int(S.indptr[b * 256]), int(S.indptr[min(n, (b + 1) * 256)])]
        out["rows_in_block"] = (stored_rows[stored_rows // 256 == b] - b * 256).tolist()[:40]
    print(json.dumps(out))
