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


def test_cpu_baseline_is_the_first_sweep_plus_measured_restart_cycles():
    # init + factorize_from(1, ncv) + restart cycles, extrapolated by the steady-state cycle — checked at a size
    # where the oracle's complete solve takes a second: the estimate must be within 2x of the solve it stands for
    import time

    import bench
    import oracle as O

    class A:
        n, nev, ncv, tol, selection, cpu_steps, cpu_cycles = 40_000, 20, 40, 1e-11, "LargestMagn", 0, 2

    rp, ci, v = O.synth_band_csr(A.n)
    # two wall-clock measurements are compared: on a machine that is busy with other work (a parallel test run) one of them can be
    # stretched, so the pair is taken up to three times and has to agree once
    ratios = []
    for _ in range(3):
        s = O.SymEigsSolver(O.Op.csr(A.n, A.n, rp, ci, v), A.nev, A.ncv)
        t0 = time.perf_counter()
        s.init()
        nconv = s.compute(O.LargestMagn, 1000, A.tol)
        full = time.perf_counter() - t0
        out = bench.cpu_baseline(A(), s.num_operations(), nconv, s.num_iterations())
        assert out["kind"] == "port" and out["cores"] == 1 and out["operations_first_sweep"] == A.ncv + 1
        assert out["operations_restart_cycles"] > 0 and "restart cycle" in out["sample"]
        ratios.append(out["estimated_seconds_per_solve"] / full)
        if 0.5 < ratios[-1] < 2.0:
            break
    assert 0.5 < ratios[-1] < 2.0, ratios


def test_pmc_summary_of_committed_counter_files():
    # what bench.py's live PMC passes do with the two rocprofv3 CSVs, on the committed passes of round 3: FETCH_SIZE calibrated
    # on the probe's k_scale launches (x 2 on gfx950), the fused SpMV instantiations found by name
    import importlib.util

    import bench

    spec = importlib.util.spec_from_file_location("pmc_summarize", os.path.join(ROOT, "tools", "pmc_summarize.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    d = mod.summarize(os.path.join(ROOT, "profiles", "r06j_pmc_fetch_counter_collection.csv"),
                      os.path.join(ROOT, "profiles", "r06j_pmc_write_counter_collection.csv"), 10_000_000)
    assert d["calibration"]["found"] and abs(d["calibration"]["read"] - 2.0) < 0.01 and abs(d["calibration"]["write"] - 1.0) < 0.01
    post = bench.fused_spmv_bytes(d["kernels"], 2, True)
    plain = bench.fused_spmv_bytes(d["kernels"], 2, False)
    assert abs(post / 1.52e9 - 1.0) < 0.02 and abs(plain / 1.52e9 - 1.0) < 0.02  # the bytes the kernel has to move, to 2 %
    assert bench.fused_spmv_bytes(d["kernels"], 0, False) is None             # that probe ran the diagonal storage only


def test_live_pmc_falls_back_without_a_profiler(monkeypatch):
    # no rocprofv3 / a failing pass must leave the committed figure in place, never raise
    import shutil

    import bench

    monkeypatch.setattr(shutil, "which", lambda name: None)
    monkeypatch.setattr(os.path, "exists", lambda p: False if str(p).endswith("rocprofv3") else True)
    value, note = bench.live_pmc_traffic(10_000_000, 2, True, timeout=5)
    assert value is None and "rocprofv3" in note


def test_live_pmc_reports_a_failed_pass():
    # in this container rocprofv3 exists but there is no GPU: the probe exits non-zero, the function says which pass failed
    import shutil

    import bench

    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        import pytest

        pytest.skip("no rocprofv3 here")
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("a GPU is present: the passes would succeed")
    value, note = bench.live_pmc_traffic(1_000_000, 2, True, timeout=120)
    assert value is None and "pass" in note


def test_flat_scalars_the_driver_keeps():
    # The driver's BENCH_rNN.json keeps only the scalars of the `roofline` block: the int32 CSR kernel's in-loop figure on SURVEY
    # 8d's bytes and the secondary workloads must be there as flat scalars.  Replayed on the committed final-tree record with the
    # flat keys stripped: flatten_for_the_driver must put every one of them back, and csr_kernel_frac must be
    # (12 nnz + 20 n + 4) bytes / time / 8 TB/s.
    import json

    import bench

    with open(os.path.join(ROOT, "profiles", "r10f_bench_final_tree.json")) as f:
        rec = json.load(f)
    want = {k: v for k, v in rec["roofline"].items() if k.startswith(("csr_kernel_", "secondary_")) or k in ("orth_frac", "host_syncs_per_solve")}
    assert {"csr_kernel_frac", "csr_kernel_ms", "secondary_m_rand_frac", "secondary_jitter_band_frac", "secondary_stencil_rcm_frac",
            "secondary_jitter_band_format", "secondary_c4_seconds", "secondary_c5_solve_ms", "orth_frac", "host_syncs_per_solve"} <= set(want)
    stripped = json.loads(json.dumps(rec))
    for k in want:
        del stripped["roofline"][k]
    bench.flatten_for_the_driver(stripped)
    assert {k: stripped["roofline"][k] for k in want} == want
    assert all(not isinstance(v, (dict, list)) for k, v in stripped["roofline"].items() if k in want)
    n, nnz = rec["config"]["n"], rec["config"]["nnz_per_gpu"]
    frac = (12.0 * nnz + 20.0 * n + 4.0) / (want["csr_kernel_ms"] * 1e-3) / 8e12
    assert abs(frac - want["csr_kernel_frac"]) <= 1e-9 and want["secondary_jitter_band_format"] == 0 and want["secondary_stencil_rcm_format"] == 0
