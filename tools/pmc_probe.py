"""Small workload for rocprofv3 --pmc passes (HBM traffic of the hot kernels at BASELINE.json's size).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -o fetch -- python tools/pmc_probe.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT -o write -- python tools/pmc_probe.py

For each storage format of the matrix (diagonal, offset-coded CSR, int32 CSR), in order: k_scale_step launches on an n-vector (calibration: exactly 8n bytes read + 8n written with the same
16-byte-per-lane access pattern the guide's FETCH_SIZE correction is about), 5 x stand-alone SpMV, then
init() + one full Lanczos factorisation (39 steps: fused SpMV, RESID_VTF, CORRECT_VTF) and one restart
(shifted-QR kernel + V*Q), in the reference flow and again with the one-sweep steps (post-scaled SpMV, k_orth_lagged).  tools/pmc_summarize.py turns the two CSVs into per-kernel bytes per launch.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import spectra_amd as sa

n = int(os.environ.get("PROBE_N", 10_000_000))
ctx = sa.default_context()
op = sa.SparseSymMatProd.synth_band(n, ctx=ctx)
x = torch.rand(n, dtype=torch.float64, device="cuda") - 0.5
y = torch.empty(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
# every storage format of the matrix in one process, so that one pair of --pmc passes covers every SpMV instantiation
# (PROBE_FORMATS=2,1,0: diagonal storage, offset-coded CSR, int32 CSR)
for fmt in [int(f) for f in os.environ.get("PROBE_FORMATS", "2,1,0").split(",")]:
    op.set_spmv_format(fmt)
    for _ in range(int(os.environ.get("PROBE_SPMV_REPS", 5))):
        op.spmv_device(x.data_ptr(), y.data_ptr())
    ctx.sync()
    # the reference flow (k_scale_step — the calibration kernel —, fused SpMV, RESID_VTF, CORRECT_VTF), then the one-sweep steps
    # (post-scaled SpMV on diagonal storage, k_orth_lagged)
    # PROBE_MODES=onesweep: only that flow (the fused SpMV instantiation <EPI, NG, NCW, false> is launched by the reference flow
    # — with the v_prev operand — AND by the one-reduction steps of the one-sweep flow — without it: one flow per pass keeps
    # the per-kernel traffic unmixed)
    for mode in os.environ.get("PROBE_MODES", "reference,onesweep").split(","):
        fac = sa.Factorization(op, 40, True)
        fac.set_orth_mode(mode)
        fac.init_random(0)
        fac.factorize_from(1, 40)
        ev, U = fac.tridiag_eigen()
        order = np.argsort(-np.abs(ev))
        fac.restart_sym(ev[order][25:])
        fac.factorize_from(25, 40)
        ctx.sync()
        print("probe done: format", op.spmv_format(), mode, "k =", fac.subspace_dim(), "nops =", fac.num_operations())
        del fac
op.set_spmv_format(-1)
