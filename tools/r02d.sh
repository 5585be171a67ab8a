OUT=gpurun_out/r02d; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_shift.py -m gpu -q -k "zero_leading" > $OUT/pytest_zero.log 2>&1; grep -E "^E|passed|failed" $OUT/pytest_zero.log | tail -5
for s in 0 2 4; do MISPEC_TILES_SYNC=$s timeout 300 python tools/bench_mrand.py 1e7 >> $OUT/mrand.jsonl 2>> $OUT/mrand.err; done
MISPEC_TILES_SYNC=2 MISPEC_TILES_WG_PER_CU=3 timeout 300 python tools/bench_mrand.py 1e7 >> $OUT/mrand.jsonl 2>> $OUT/mrand.err
cat $OUT/mrand.jsonl
MISPEC_TILES_SYNC=2 bash tools/pmc_pass.sh $OUT l2_sync2 "TCC_HIT_sum TCC_MISS_sum" tools/pmc_probe_mrand.py
python - <<'PY' > $OUT/c5_chunks.jsonl 2>&1
import json, os, subprocess, sys
code = '''
import sys, time, json, numpy as np, scipy.sparse as sp
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import spectra_amd as sa
from test_gpu_fullsize import banded_spd
n = 2_000_000
A = banded_spd(n, 3, seed=5)
op = sa.SparseSymShiftSolve(sp.tril(A).tocsc())
op.set_shift(0.0)
t0 = time.perf_counter(); op.set_shift(0.0); tf = time.perf_counter() - t0
best = None
for r in range(3):
    s = sa.SymEigsShiftSolver(op, 6, 20, 0.0); s.profile(2)
    t0 = time.perf_counter(); s.init(); nc = s.compute(sa.SortRule.LargestMagn, 1000, 1e-11); op.ctx.sync(); dt = time.perf_counter() - t0
    p = s.get_profile()
    if r and (best is None or dt < best[0]): best = (dt, p["ms_spmv"] / p["n_spmv"], nc, s.num_operations())
x = np.random.default_rng(0).uniform(-1, 1, n); y = op.perform_op(x)
res = np.abs(A @ y - x).max()
print(json.dumps({"chunks": __import__("os").environ.get("MISPEC_SHIFT_CHUNK", "default"), "set_shift_s": round(tf, 4), "solve_ms": round(best[1], 4), "solve_s": round(best[0], 4), "nconv": best[2], "nops": best[3], "resid": res, "info": op.refinement_info()}))
'''
for cfg in ("", "64,32,500000,4096", "32,32,0,4096", "64,64,0,4096", "48,24,500000,4096"):
    env = dict(os.environ)
    if cfg: env["MISPEC_SHIFT_CHUNK"] = cfg
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    print(r.stdout.strip() or json.dumps({"chunks": cfg, "error": r.stderr[-400:]}), flush=True)
PY
cat $OUT/c5_chunks.jsonl
