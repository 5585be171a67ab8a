"""What ONE rank of an N-way row-sharded C3 run executes, measured on one GPU: the sharded code path (communicator attached:
loopback world = 1, so every collective is issued and lands on the rank itself; one speculative correction per step and the
restart sweeps on the host, as with world > 1) on an M-band matrix of the SHARD's size, k = 20, ncv = 40, tol 1e-11.

    python tools/shard_profile.py ROWS [reference|onesweep] [solves]        # prints one JSON line
    rocprofv3 --kernel-trace --stats ... -- python tools/shard_profile.py 1250000 onesweep   # per-kernel split (tools/gpu_call.sh trace)

The wire time of the collectives is NOT in these numbers (a 1-GPU box cannot measure it); everything else a rank does is."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MISPEC_SPEC_CORR", "1")
os.environ.setdefault("MISPEC_SMALL", "host")

import spectra_amd as sa  # noqa: E402
from spectra_amd import _capi  # noqa: E402


def main():
    rows = int(sys.argv[1])
    orth = sys.argv[2] if len(sys.argv) > 2 else "reference"
    solves = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    lib = sa.lib()
    grp = C.c_void_p()
    _capi.check(lib.mispec_loopback_create(1, C.byref(grp)))
    ctx = sa.Context(0)
    _capi.check(lib.mispec_loopback_attach(grp, ctx.h, 0))
    ctx.rank, ctx.world = 0, 1
    op = sa.SparseSymMatProd.synth_band(rows, ctx=ctx)
    eigs = sa.SymEigsSolver(op, 20, 40)
    eigs.set_orth_mode(orth)
    def solve():
        ctx.sync()
        t0 = time.perf_counter()
        eigs.init()
        nconv = eigs.compute(sa.SortRule.LargestMagn, 1000, 1e-11)
        eigs.eigenvectors(to_host=False)
        ctx.sync()
        return nconv, time.perf_counter() - t0

    for _ in range(max(1, solves - 1)):  # warm-up, then the wall-clock figure without any event records in the stream
        nconv, dt = solve()
    nops = eigs.num_operations()
    eigs.profile(1)
    p0 = eigs.get_profile()
    solve()  # one more with every kernel family bracketed by HIP events: the split
    p1 = eigs.get_profile()
    out = {"rows_per_rank": rows, "orth": orth, "nconv": int(nconv), "num_operations": int(nops), "num_iterations": int(eigs.num_iterations()),
           "seconds_per_solve": dt, "ms_per_operator_application_all_inclusive": 1e3 * dt / nops,
           "kernel_families_ms_per_operation": {k[3:]: (p1[k] - p0[k]) / nops for k in p1 if k.startswith("ms_")},
           "launches_per_operation": {k[2:]: (p1[k] - p0[k]) / nops for k in p1 if k.startswith("n_")},
           "max_residual": float(eigs.residuals().max()), "orth_info": eigs.orth_info(), "turn_info": eigs.turn_info(),
           "options": {k: sa.get_option(k) for k in ("host_turn", "orth_kernel", "small", "spec_corr")},
           "note": "wall-clock figure from an un-instrumented solve; the family split from one more solve with HIP events; loopback world = 1"}
    print(json.dumps(out), flush=True)
    del eigs, op
    _capi.check(lib.mispec_loopback_destroy(grp))


if __name__ == "__main__":
    main()
