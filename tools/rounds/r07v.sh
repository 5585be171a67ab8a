#!/bin/bash
# round 4, GPU call: one-sweep steps on bases of up to 128 columns
OUT=gpurun_out/r07v; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_onesweep.py tests/test_gpu_sharded.py tests/test_gpu_solver.py > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
python - <<"PY" > $OUT/wide_timing2.jsonl 2>&1
import json, time
import spectra_amd as sa
ctx = sa.default_context()
op = sa.SparseSymMatProd.synth_band(10_000_000, ctx=ctx)
for nev, ncv in [(50, 100), (60, 128), (30, 64)]:
    for mode in ("reference", "onesweep"):
        e = sa.SymEigsSolver(op, nev, ncv); e.set_orth_mode(mode); e.profile(1)
        ctx.sync(); t0 = time.perf_counter(); e.init(); nconv = e.compute(sa.SortRule.LargestMagn, 6, 1e-11); ctx.sync(); dt = time.perf_counter() - t0
        p = e.get_profile()
        print(json.dumps({"n": 10_000_000, "nev": nev, "ncv": ncv, "orth": mode, "restarts": 6, "seconds": dt, "num_operations": int(e.num_operations()),
                          "ms_per_operation": 1e3 * dt / e.num_operations(), "kernels_ms": {k[3:]: round(v, 1) for k, v in p.items() if k.startswith("ms_")}}), flush=True)
        del e
PY
cat $OUT/wide_timing.jsonl | cut -c1-400
