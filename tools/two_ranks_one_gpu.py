"""Can the library's RCCL transport run with two ranks on ONE device?  (the test tier has one GPU.)

    python tools/two_ranks_one_gpu.py [out.json]

Starts two ranks through torch.distributed.run, both on device 0, MP_TRANSPORT=rccl (tests/mp_worker.py), and records what
happens: RCCL normally refuses a communicator with duplicate devices ("Duplicate GPU detected"), in which case the
multi-process tests use the gloo-staged transport instead.  Prints / writes one JSON object."""
import json
import os
import socket
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
out_dir = tempfile.mkdtemp()
env = dict(os.environ, MP_OUT=out_dir, MP_TRANSPORT="rccl", MP_DEVICE="shared", NCCL_DEBUG="WARN")
cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
       os.path.join(ROOT, "tests", "mp_worker.py")]
try:
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    rc, tail = r.returncode, r.stdout[-3000:]
except subprocess.TimeoutExpired as e:
    rc, tail = -9, ((e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""))[-3000:]
res = {"what": "two ranks on one device through the RCCL transport (nccl process group + mispec_ctx_set_comm_rccl)", "returncode": rc,
       "worked": rc == 0 and os.path.exists(os.path.join(out_dir, "rank1.npz")), "output_tail": tail}
print(json.dumps(res))
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        json.dump(res, f, indent=1)
