"""Latency of SMALL solves (BASELINE.json configs[0]: the reference's own 1000 x 1000 fixture, and a few sizes above it): seconds
per solve and per operation on the GPU against the CPU oracle on the same matrix — the regime where launches, not bytes, bound
the device.      python tools/small_latency.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import scipy.sparse as sp

import oracle as O
import spectra_amd as sa
from helpers import sparse_fixture

ctx = sa.default_context()
for n, prob, k, m in [(1000, 0.01, 20, 50), (10_000, 0.001, 20, 50), (100_000, None, 20, 40), (1_000_000, None, 20, 40)]:
    if prob is None:
        op = sa.SparseSymMatProd.synth_band(n, ctx=ctx)
        rp, ci, v = O.synth_band_csr(n)
        oop = O.Op.csr(n, n, rp, ci, v)
    else:
        A, S = sparse_fixture(n, prob)
        op = sa.SparseSymMatProd(A, ctx=ctx)
        Sc = sp.csr_matrix(S)
        Sc.sort_indices()
        oop = O.Op.csr(n, n, Sc.indptr.astype(np.int32), Sc.indices.astype(np.int32), Sc.data)
    for mode in ("onesweep", "reference"):
        best = None
        for rep in range(4):
            e = sa.SymEigsSolver(op, k, m)
            e.set_orth_mode(mode)
            ctx.sync()
            t0 = time.perf_counter()
            e.init()
            nconv = e.compute(sa.SortRule.LargestAlge, 1000, 1e-10)
            X = e.eigenvectors()
            dt = time.perf_counter() - t0
            if rep and (best is None or dt < best[0]):
                best = (dt, nconv, e.num_operations(), e.num_iterations(), e.get_profile()["n_host_sync"] if "n_host_sync" in e.get_profile() else None)
        t0 = time.perf_counter()
        o = O.SymEigsSolver(oop, k, m)
        o.init()
        onconv = o.compute(O.LargestAlge, 1000, 1e-10)
        oX = o.eigenvectors()
        odt = time.perf_counter() - t0
        print(json.dumps({"n": n, "nev": k, "ncv": m, "orth": mode, "gpu_seconds": best[0], "gpu_us_per_operation": 1e6 * best[0] / best[2],
                          "nconv": int(best[1]), "num_operations": int(best[2]), "num_iterations": int(best[3]), "host_syncs": best[4],
                          "cpu_oracle_seconds_1_thread": odt, "cpu_num_operations": int(o.num_operations()),
                          "gpu_over_cpu": odt / best[0]}), flush=True)
