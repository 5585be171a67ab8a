"""C4 (GenEigsSolver 5M, k=10, ncv=30): how often does the DGKS test (Arnoldi.h:257) ask for a re-orthogonalisation, and where does the time go?"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectra_amd as sa
ctx = sa.default_context()
gop = sa.SparseGenMatProd.synth_band(5_000_000, ctx=ctx)
for level in (0, 1):
    g = sa.GenEigsSolver(gop, 10, 30)
    if level:
        g.profile(level)
    for rep in range(2):
        ctx.sync(); t0 = time.perf_counter()
        g.init(); nconv = g.compute(sa.SortRule.LargestMagn, 1000, 1e-11); ctx.sync()
        dt = time.perf_counter() - t0
    out = {"profile_level": level, "seconds": dt, "nconv": int(nconv), "num_operations": int(g.num_operations()), "num_iterations": int(g.num_iterations())}
    if level:
        p = g.get_profile()
        out["profile"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in p.items()}
    print(json.dumps(out), flush=True)
