"""Same-box A/B of environment switches on the headline workload: runs `bench.py --no-secondary --no-cpu-baseline` once per
variant (each in its own process: the switches are read at library / matrix construction) and prints one line per variant.

    python tools/ab_bench.py [--steps K] NAME=ENV1=v,ENV2=v ...     e.g.  base= dia_diagonal=MISPEC_DIA_LAYOUT=diagonal
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = "3"
args = sys.argv[1:]
if args and args[0] == "--steps":
    steps = args[1]
    args = args[2:]
for spec in args:
    name, _, envs = spec.partition("=")
    env = dict(os.environ)
    for kv in filter(None, envs.split(",")):
        k, _, v = kv.partition("=")
        env[k] = v
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "1", "--no-secondary", "--no-cpu-baseline"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        k = d.get("kernels_ms_per_solve") or {}
        print(json.dumps({"variant": name, "env": envs, "eigenpairs_per_s": round(d["value"], 3), "ms_per_solve": round(d["ms_per_step"], 1),
                          "spmv_ms_per_launch": round(d["roofline"]["ms_per_launch"], 4), "spmv_frac": round(d["roofline"]["frac"], 4),
                          "spmv_standalone_ms": round(d["roofline"]["standalone_ms_per_launch"], 4),
                          "kernels_ms_per_solve": {a: round(b, 1) for a, b in k.items()}, "max_residual": d["solve"]["max_residual"],
                          "num_operations": d["solve"]["num_operations"]}), flush=True)
    except Exception as e:  # noqa: BLE001
        print(json.dumps({"variant": name, "error": repr(e), "stderr": r.stderr[-500:]}), flush=True)
