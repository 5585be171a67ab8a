import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import spectra_amd as sa
ctx = sa.default_context()
for n, offs, nev, ncv, rule in [(2001, (1, 2, 3), 10, 64, "LargestAlge"), (2001, (1, 2, 3), 10, 64, "BothEnds"), (5001, (1, 7, 300), 12, 48, "LargestMagn"),
                                (5001, (1, 7, 300), 6, 64, "SmallestAlge"), (20001, (1, 2, 3, 50, 51, 1500, 1501), 10, 60, "LargestAlge"),
                                (50_001, (1, 2, 3, 100, 101, 5000, 5001), 8, 24, "LargestAlge"), (1001, (1,), 5, 40, "LargestAlge"), (3001, (1, 2), 20, 64, "BothEnds")]:
    op = sa.SparseSymMatProd.synth_band(n, offsets=offs, ctx=ctx)
    for mode in ("onesweep", "onesweep-eager"):
        e = sa.SymEigsSolver(op, nev, ncv)
        e.set_orth_mode(mode)
        e.init()
        nconv = e.compute(sa.SortRule[rule], 1000, 1e-11)
        oi = e.orth_info()
        print(n, offs, nev, ncv, rule, mode, "nconv", nconv, "ops", e.num_operations(), "it", e.num_iterations(), {k: oi[k] for k in ("lagged_steps", "check_stops", "state_stops", "fused_restarts", "fused_redone")}, "res %.2e" % e.residuals().max())
