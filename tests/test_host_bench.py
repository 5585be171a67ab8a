"""Host-side logic of bench.py that needs no GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_bench_spawns_its_own_ranks():
    # `python bench.py --gpus 2` started as a plain process must start two ranks itself (re-exec through
    # torch.distributed.run); on a 1-GPU box the second rank has no device, so only the spawn mechanics are checked here:
    # the child command line and the environment hand-over.
    import bench

    class A:
        gpus = 2

    calls = {}
    orig = subprocess.call
    try:
        subprocess.call = lambda cmd, env=None: calls.update(cmd=cmd, env=env) or 0
        sys_argv = sys.argv
        sys.argv = ["bench.py", "--gpus", "2", "--steps", "1"]
        assert bench.respawn_as_ranks(A()) == 0
    finally:
        subprocess.call = orig
        sys.argv = sys_argv
    cmd = calls["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "2", "--steps", "1"]
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ


def test_pmc_traffic_lookup_finds_every_instantiation():
    # `roofline.traffic` must never be null for the headline size: a committed PMC summary exists for each SpMV format
    import bench

    for fmt in (0, 1, 2):
        traffic, src = bench.pmc_traffic(10_000_000, fmt)
        assert traffic is not None and 1.0e9 < traffic < 3.0e9, (fmt, traffic, src)
    # ... and for the post-scaled instantiation the one-sweep steps run on diagonal storage (bench.py's timed mode)
    traffic, src = bench.pmc_traffic(10_000_000, 2, post_scaled=True)
    assert traffic is not None and 1.4e9 < traffic < 1.7e9, (traffic, src)
