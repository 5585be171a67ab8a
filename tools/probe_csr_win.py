"""int32 CSR kernel with x windows in LDS (k_spmv_csr_win) against the gather kernel (k_spmv_csr_stream): bit-identity of the products
and stand-alone / in-loop times, on M-band, the jittered band and the 7-point stencil.  One JSON line per (matrix, variant).
    python tools/probe_csr_win.py [n] [matrices: band,jitter,stencil]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MISPEC_KERNEL_PROBE"] = "1"  # the library re-reads the kernel knobs at every launch (csr.hip launch_spmv_raw)
import numpy as np
import scipy.sparse as sp
import torch

import spectra_amd as sa
from spectra_amd import workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["band", "jitter", "stencil"]
ctx = sa.default_context()
VARIANTS = [("gather", None), ("win i2 pf0", ("2", "0", "0")), ("win i2 pf0 nt", ("2", "0", "1")), ("win i2 pf1 nt", ("2", "1", "1")),
            ("win i1 pf0", ("1", "0", "0")), ("win i1 pf1", ("1", "1", "0")), ("win i1 pf0 nt", ("1", "0", "1")), ("win i1 pf1 nt", ("1", "1", "1"))]
if os.environ.get("PROBE_VARIANTS"):
    VARIANTS = [v for v in VARIANTS if v[0] in os.environ["PROBE_VARIANTS"].split(";")]


def set_variant(op, v):
    op.use_windows(v is not None)
    for k in ("MISPEC_CSR_WIN_ITERS", "MISPEC_CSR_WIN_PF", "MISPEC_CSR_WIN_NT"):
        os.environ.pop(k, None)
    if v:
        os.environ["MISPEC_CSR_WIN_ITERS"], os.environ["MISPEC_CSR_WIN_PF"], os.environ["MISPEC_CSR_WIN_NT"] = v


def run(name, op, nev=20, ncv=40, restarts=10):
    nr = op.rows()
    x = torch.rand(nr, dtype=torch.float64, device="cuda") - 0.5
    y = torch.empty(nr + 2, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()  # x is written on torch's stream, the library runs on its own
    auto = op.spmv_format()
    info = op.windows_info()
    ref = None
    if auto == 2:  # the diagonal format is pinned to the oracle at full size (tests/test_gpu_fullsize.py)
        op.spmv_device(x.data_ptr(), y.data_ptr())
        ctx.sync()
        ref = y[:nr].clone()
    op.set_spmv_format(0)
    for vname, v in VARIANTS:
        set_variant(op, v)
        y.zero_()
        op.spmv_device(x.data_ptr(), y.data_ptr())
        ctx.sync()
        if ref is None:
            ref = y[:nr].clone()
        same = bool(torch.equal(ref, y[:nr]))
        op.spmv_time(x.data_ptr(), y.data_ptr(), 3)
        alone = op.spmv_time(x.data_ptr(), y.data_ptr(), 20)
        e = sa.SymEigsSolver(op, nev, ncv)
        e.profile(2)
        e.init()
        t0 = time.perf_counter()
        nconv = e.compute(sa.SortRule.LargestMagn, restarts, 1e-11)
        ctx.sync()
        dt = time.perf_counter() - t0
        p = e.get_profile()
        inloop = p["ms_spmv"] / max(p["n_spmv"], 1)
        alg = op.algorithmic_bytes()
        H = np.array(e.ritz_values()) if hasattr(e, "ritz_values") else None
        print(json.dumps({"matrix": name, "variant": vname, "n": nr, "nnz": op.nnz(), "auto_format": auto, "windows": info,
                          "bit_identical": same, "standalone_ms": round(alone, 4), "standalone_frac_8d": round(alg / (alone * 1e-3) / 8e12, 4),
                          "in_loop_ms": round(inloop, 4), "in_loop_frac_8d": round(alg / (inloop * 1e-3) / 8e12, 4),
                          "in_loop_launches": int(p["n_spmv"]), "solve_s": round(dt, 3), "num_operations": int(e.num_operations())}), flush=True)
        del e
    set_variant(op, ("2", "1", "0"))
    op.use_windows(None)
    for k in ("MISPEC_CSR_WIN_ITERS", "MISPEC_CSR_WIN_PF", "MISPEC_CSR_WIN_NT"):
        os.environ.pop(k, None)
    op.set_spmv_format(-1)


if "band" in which:
    op = sa.SparseSymMatProd.synth_band(n, ctx=ctx)
    run("M-band", op)
    del op
if "jitter" in which:
    t0 = time.perf_counter()
    A = workloads.jitter_band(n)
    tg = time.perf_counter() - t0
    op = sa.SparseSymMatProd(sp.tril(A).tocsc(), ctx=ctx)
    print(json.dumps({"matrix": "jitter band", "host_generation_s": round(tg, 1), "ingest": sa.last_ingest_info()}), flush=True)
    # product against scipy on a sample of rows (the gather kernel itself is pinned to the oracle elsewhere)
    run("jitter band", op)
    del op, A
if "stencil" in which:
    m = int(round(n ** (1.0 / 3.0)))
    A = workloads.stencil7(m)
    op = sa.SparseSymMatProd(sp.tril(A).tocsc(), ctx=ctx)
    run("stencil7 natural order m=%d" % m, op, 10, 30, 8)
    del op
    perm = np.random.default_rng(1).permutation(A.shape[0])
    B = A[perm][:, perm].tocsr()
    B.sort_indices()
    del A
    op = sa.SparseSymMatProd(sp.tril(B).tocsc(), ctx=ctx)
    print(json.dumps({"matrix": "stencil7 random order", "reordering": op.reordering_info(), "format": op.spmv_format()}), flush=True)
    run("stencil7 random order -> RCM at ingest", op, 10, 30, 8)
