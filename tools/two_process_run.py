"""A row-sharded solve as TWO operating-system processes on this box (one GPU: the ranks share device 0, collectives staged
through host memory over gloo — spectra_amd.dist.HostStagedComm), compared with the single-process solve of the same problem.

    python tools/two_process_run.py [out.json]

Records: process ids, exchange plan, overlap plan, eigenvalues of both runs, their difference, operation counts, wall times.
(RCCL cannot build a communicator with two ranks on one device: tools/two_ranks_one_gpu.py records that.)"""
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import spectra_amd as sa

n, offsets, nev, ncv = 2_000_000, (1, 2, 3, 1000, 1001, 100000, 100001), 8, 24
res = {"n": n, "offsets": offsets, "nev": nev, "ncv": ncv, "runs": {}}
for exchange in (None, "allgather"):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out_dir = tempfile.mkdtemp()
    env = dict(os.environ, MP_OUT=out_dir, MP_TRANSPORT="gloo-staged", MP_DEVICE="shared", MP_N=str(n), MP_OFFSETS=",".join(map(str, offsets)),
               MP_NEV=str(nev), MP_NCV=str(ncv))
    env.pop("MISPEC_EXCHANGE", None)
    if exchange:
        env["MISPEC_EXCHANGE"] = exchange
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "mp_worker.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    wall = time.perf_counter() - t0
    rec = {"returncode": r.returncode, "wall_seconds_incl_startup": wall}
    if r.returncode == 0:
        ranks = [dict(np.load(os.path.join(out_dir, f"rank{k}.npz"))) for k in range(2)]
        metas = [json.load(open(os.path.join(out_dir, f"rank{k}.json"))) for k in range(2)]
        rec.update({"pids": [m["pid"] for m in metas], "halo": [m["halo"] for m in metas], "nconv": [int(x["nconv"]) for x in ranks],
                    "num_operations": [int(x["nops"]) for x in ranks], "eigenvalues_rank0": [float(v) for v in ranks[0]["evals"]],
                    "ranks_hold_identical_eigenvalues": bool(np.array_equal(ranks[0]["evals"], ranks[1]["evals"])),
                    "doubles_received_per_product": [int(x["exchange"][1]) for x in ranks], "max_residual": float(max(x["res"].max() for x in ranks)),
                    "rows": [[int(a) for a in x["rows"]] for x in ranks]})
    else:
        rec["output_tail"] = r.stdout[-1500:]
    res["runs"]["halo" if exchange is None else "allgather"] = rec
single = sa.SymEigsSolver(sa.SparseSymMatProd.synth_band(n, offsets=offsets), nev, ncv)
single.init()
nconv = single.compute(sa.SortRule.LargestMagn, 1000, 1e-11)
ev = single.eigenvalues()
res["single_process"] = {"nconv": int(nconv), "num_operations": int(single.num_operations()), "eigenvalues": [float(v) for v in ev]}
for k, rec in res["runs"].items():
    if "eigenvalues_rank0" in rec:
        rec["max_abs_diff_vs_single_process"] = float(np.abs(np.array(rec["eigenvalues_rank0"]) - ev).max())
print(json.dumps(res))
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        json.dump(res, f, indent=1)
