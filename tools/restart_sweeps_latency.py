"""Latency of the m x m work of one restart, every variant: the shifted QR sweeps on the host in the
reference's serial order, as the skewed pipeline (host, device kernel), the one-wavefront kernel; plus the host's TridiagEigen
(Ritz values + last row) from tools/probes/host_small_latency.cpp when a compiler is there.

    python tools/restart_sweeps_latency.py [m] [shifts]
"""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import spectra_amd as sa

m = int(sys.argv[1]) if len(sys.argv) > 1 else 40
p = int(sys.argv[2]) if len(sys.argv) > 2 else 18
rng = np.random.default_rng(1)
d = 4.0 * (rng.random(m) - 0.5)
e = 1.0 + (rng.random(m - 1) - 0.5)
mu = 3.0 * (rng.random(p) - 0.5)
ctx = sa.default_context()
out = {"m": m, "shifts": p}
ref = sa.restart_sweeps(d, e, mu, "host-serial")
for v in ("host-serial", "host-pipelined", "device-pipelined", "device-wavefront"):
    r = None
    best = 1e30
    for _ in range(5):
        r = sa.restart_sweeps(d, e, mu, v, reps=200, ctx=ctx)
        best = min(best, r[3])
    out[v + "_us"] = round(best, 2)
    out[v + "_bit_identical_to_serial"] = bool(all(np.array_equal(x, y) for x, y in zip(ref[:3], r[:3])))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
try:
    exe = "/tmp/host_small_latency"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(root, "include"), os.path.join(root, "tools", "probes", "host_small_latency.cpp"), "-o", exe])
    out["host_probe"] = subprocess.check_output([exe]).decode().strip()
    out["cpu"] = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
except Exception as ex:  # noqa: BLE001
    out["host_probe"] = repr(ex)
print(json.dumps(out))
