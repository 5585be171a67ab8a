"""Same-box A/B of environment settings on the C2 solve (one child process per setting, interleaved twice).
    python tools/probes/env_ab.py NAME=VALUE[,NAME=VALUE...] ...      ("-" = no setting)"""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, json, time
sys.path.insert(0, %r)
import spectra_amd as sa
ctx = sa.default_context()
n = int(os.environ.get("PROBE_N", 10_000_000))
op = sa.SparseSymMatProd.synth_band(n, ctx=ctx)
e = sa.SymEigsSolver(op, 20, 40)
e.profile(1)
times = []
for r in range(4):
    ctx.sync(); t0 = time.perf_counter()
    e.init(); nconv = e.compute(sa.SortRule.LargestMagn, 1000, 1e-11); e.eigenvectors(to_host=False); ctx.sync()
    times.append(time.perf_counter() - t0)
p = e.get_profile()
print(json.dumps({"setting": os.environ.get("AB_SETTING"), "n": n, "seconds_min": round(min(times), 5), "seconds_all": [round(t, 4) for t in times],
                  "num_operations": int(e.num_operations()), "max_residual": float(e.residuals().max()),
                  "ms_per_solve": {k[3:]: round(v / 4, 2) for k, v in p.items() if k.startswith("ms_")}}), flush=True)
''' % ROOT
settings = sys.argv[1:] or ["-"]
for rep in range(2):
    for sset in settings:
        env = dict(os.environ, AB_SETTING=sset)
        if sset != "-":
            for kv in sset.split(","):
                k, v = kv.split("=", 1)
                env[k] = v
        subprocess.run([sys.executable, "-c", CHILD], env=env)
