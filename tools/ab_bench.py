"""Same-box A/B of environment switches on the headline workload: runs `bench.py --no-secondary --no-cpu-baseline` once per
variant (each in its own process: the switches are read at library / matrix construction) and prints one line per variant.

    python tools/ab_bench.py [--steps K] NAME=ENV1=v,ENV2=v ...     e.g.  base= host_steps=MISPEC_HOST_STEPS=1
An item that starts with "--" is passed to bench.py instead (e.g. one=--orth=onesweep).
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
    extra = []
    for kv in filter(None, envs.split(",")):
        if kv.startswith("--"):
            extra.append(kv)
            continue
        k, _, v = kv.partition("=")
        env[k] = v
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "1", "--no-secondary", "--no-cpu-baseline"] + extra,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        k = d.get("kernels_ms_per_solve") or {}
        print(json.dumps({"variant": name, "env": envs, "eigenpairs_per_s": round(d["value"], 3), "ms_per_solve": round(d["ms_per_step"], 1),
                          "spmv_ms_per_launch": round(d["roofline"]["ms_per_launch"], 4), "spmv_frac": round(d["roofline"]["frac"], 4),
                          "spmv_standalone_ms": round(d["roofline"]["standalone_ms_per_launch"], 4),
                          "kernels_ms_per_solve": {a: round(b, 1) for a, b in k.items()}, "max_residual": d["solve"]["max_residual"],
                          "num_operations": d["solve"]["num_operations"], "orth": d["solve"].get("orth_info", {}).get("mode"),
                          "other_orth_mode": (d.get("other_orth_mode") or {}).get("value"),
                          "with_host_eigenvectors": (d.get("value_with_host_eigenvectors") or {}).get("value")}), flush=True)
    except Exception as e:  # noqa: BLE001
        print(json.dumps({"variant": name, "error": repr(e), "stderr": r.stderr[-500:]}), flush=True)
