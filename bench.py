#!/usr/bin/env python
"""Headline benchmark: BASELINE.json configs[1] — SymEigsSolver on a 10M x 10M, ~15 nnz/row fp64 symmetric CSR,
k = 20, ncv = 40 — on N MI355X of one node (N > 1: the same matrix row-partitioned, the Krylov vector exchanged over
xGMI by RCCL before every SpMV = configs[2], strong scaling).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus 8                      # spawns its 8 ranks itself (re-exec through torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

A "step" is ONE complete solve: init() + compute(LargestMagn, tol) + eigenvectors() with the matrix already
resident in HBM (generated there by the counter-hash generator of SURVEY.md §8d).  Rank 0 prints one JSON line:
  value      eigenpairs/s of the whole job;
  roofline   the SpMV kernel that ran in the timed region, on the bytes THAT kernel has to move (its storage format's
             values/indices + x + y + the fused epilogue's two vector reads) / its mean launch duration from HIP events
             bound to the dispatches inside the timed region; `traffic` = the PMC-measured HBM bytes per launch of the same
             instantiation (profiles/*pmc_traffic.json); the CSR-int32 byte count of SURVEY.md 8d over the same time is
             reported separately as `csr_equivalent_gbps` (it is NOT an HBM rate when a compressed format ran);
  secondary  driver-run figures of the other configurations: CSR kernels on the same matrix, M-rand (scattered columns,
             stand-alone and inside a solver loop), C4 (GenEigsSolver 5M) and C5 (shift-and-invert 2M), all on true bytes;
  cpu_baseline  the CPU oracle (Eigen-free restatement of the reference, 1 thread like the reference) timed on a bounded
             sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
ORTH_DEFAULT = "onesweep"   # the LIBRARY'S default (round 4): what a user who only changes the include path gets; the reference-flow figure of the same run: `other_orth_mode`


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--size", dest="n", type=int, default=10_000_000, help="matrix dimension (default: BASELINE.json configs[1])")
    p.add_argument("--nev", type=int, default=20)
    p.add_argument("--ncv", type=int, default=40)
    p.add_argument("--tol", type=float, default=1e-11,
                   help="1e-11 so that ||Av - lv||/||v|| <= 1e-10 holds for |l| ~ 2.5 (the criterion is relative to |l|)")
    p.add_argument("--selection", default="LargestMagn")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-secondary", action="store_true", help="skip the secondary configurations (M-rand, C4, C5, CSR kernels)")
    p.add_argument("--no-live-pmc", action="store_true", help="do not run the two rocprofv3 --pmc passes that measure `roofline.traffic` in this run "
                   "(about a minute; the committed profiles/*pmc_traffic*.json figure is reported instead)")
    p.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    p.add_argument("--profile-level", type=int, default=2, choices=[1, 2],
                    help="HIP-event instrumentation of the timed solves: 2 = operator applications only (default), 1 = every kernel family")
    p.add_argument("--cpu-steps", type=int, default=0, help="> 0: quick CPU sample of that many Lanczos steps instead of the default (first sweep + restart cycles)")
    p.add_argument("--cpu-cycles", type=int, default=1, help="restart cycles of the CPU sample after the first sweep (SURVEY.md 8d; about 25 s each at n = 1e7)")
    p.add_argument("--spmv-reps", type=int, default=50, help="stand-alone SpMV launches timed after the solves")
    p.add_argument("--orth", default=ORTH_DEFAULT, choices=["reference", "onesweep", "onesweep-eager"],
                   help="orthogonalisation of the Lanczos steps in the timed region (include/mispec.h mispec_fac_set_orth_mode); the other "
                        "mode is timed as well, outside the timed region, and reported as `other_orth_mode`")
    return p.parse_args()


def respawn_as_ranks(args):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks through torch.distributed.run on a free
    local port and hand their output through (rank 0 prints the JSON line)."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def cpu_baseline(args, gpu_nops, gpu_nconv, gpu_niter):
    """The oracle (a 'port': Eigen is absent, see oracle/spectra_oracle.hpp — pinned bit for bit to the reference's own headers by
    tests/test_ref_pin.py) on the host, 1 thread.  SURVEY.md 8(d)'s sample: init() + factorize_from(1, ncv) (the first sweep),
    then --cpu-cycles measured restart cycles of the IRLM driver (shifts, compress_V, factorize_from(k, ncv)); a complete solve
    is the first sweep plus as many cycles as the solve makes, so the estimate is
        first sweep + (operations of the GPU run - operations of the first sweep) x seconds per operation of the measured cycles."""
    import oracle as O

    t0 = time.time()
    rp, ci, v = O.synth_band_csr(args.n)
    op = O.Op.csr(args.n, args.n, rp, ci, v)
    t_gen = time.time() - t0
    if args.cpu_steps > 0:  # quick variant for tests and hand runs: the first few steps only (cheap steps: few basis columns)
        secs, nops = O.time_lanczos_steps(op, args.ncv, args.cpu_steps)
        t_first, ops_first, t_cyc, ops_cyc, ncyc = secs, nops, 0.0, 0, 0
        per_op = secs / nops
        est_total = per_op * gpu_nops
        sample = (f"QUICK SAMPLE (--cpu-steps): oracle init() + {args.cpu_steps} Lanczos steps at n={args.n} ({nops} perform_op, "
                  f"{secs:.1f} s); early steps orthogonalise against few columns, so this flatters the CPU")
    else:
        t_first, ops_first, t_cyc, ops_cyc, ncyc = O.time_restart_cycles(op, args.nev, args.ncv, getattr(O, args.selection), args.tol, args.cpu_cycles)
        per_op = t_cyc / ops_cyc if ops_cyc else t_first / ops_first
        est_total = t_first + max(0, gpu_nops - ops_first) * per_op
        sample = (f"oracle init() + factorize_from(1, {args.ncv}) ({ops_first} perform_op, {t_first:.1f} s) + {ncyc} restart cycle(s) "
                  f"(shifted QR, compress_V, factorize_from(k, {args.ncv}): {ops_cyc} perform_op, {t_cyc:.1f} s = {per_op:.3f} s/op in "
                  f"steady state) at n={args.n}; complete solve estimated as first sweep + ({gpu_nops} - {ops_first}) x {per_op:.3f} s "
                  f"= {est_total:.0f} s; matrix generation {t_gen:.1f} s not timed")
    x = O.simple_random(args.n, 0)
    t_spmv = op.time_op(x, 3)
    out = {
        "value": gpu_nconv / est_total,
        "unit": "eigenpairs/s",
        "cores": 1,
        "kind": "port",
        "sample": sample,
        "seconds_first_sweep": t_first,
        "operations_first_sweep": ops_first,
        "seconds_restart_cycles": t_cyc,
        "operations_restart_cycles": ops_cyc,
        "seconds_per_op_steady_state": per_op,
        "estimated_seconds_per_solve": est_total,
        "spmv_seconds": t_spmv,
        "spmv_gbps": (12.0 * len(v) + 20.0 * args.n + 4) / t_spmv / 1e9,
    }
    # the oracle's COMPLETE solve of this configuration, run once on the build host (tests/golden/full_size_c2.json)
    try:
        with open(os.path.join(ROOT, "tests", "golden", "full_size_c2.json")) as f:
            g = json.load(f)
        if g["n"] == args.n and g["nev"] == args.nev and g["ncv"] == args.ncv:
            ratio = g["oracle_seconds_one_thread"] / est_total
            out["complete_solve_on_build_host"] = {
                "seconds": g["oracle_seconds_one_thread"], "num_operations": g["num_operations"],
                "eigenpairs_per_s": g["nconv"] / g["oracle_seconds_one_thread"],
                "ratio_to_this_estimate": ratio,
                "build_host_oracle_spmv_gbps": g.get("build_host_oracle_spmv_gbps"),
                "note": ("not this box: the oracle's complete solve as recorded on the build container's host, whose one core runs the "
                         f"oracle's SpMV at {g.get('build_host_oracle_spmv_gbps')} GB/s against {out['spmv_gbps']:.1f} GB/s here — the same code and "
                         f"the same {g['num_operations']} operations; the ratio of the two solve times ({ratio:.1f}) is the ratio of the two hosts' "
                         "memory systems (every kernel of the solve is a stream), not of the sample's extrapolation")}
    except Exception:  # noqa: BLE001 - informational only
        pass
    return out


KERNEL_OF_FORMAT = {0: "k_spmv_csr_win (int32 CSR, x windows in LDS) / k_spmv_csr_stream<EPI, NT, 256, CODES=false> (gathers)", 1: "k_spmv_csr_stream<EPI, NT, 256, CODES=true>", 2: "k_spmv_dia_win2 (two rows per thread, 16-byte loads) / k_spmv_dia_win / k_spmv_dia",
                    3: "k_spmv_tiles (column-blocked tiles, segment sums in LDS)",
                    4: "k_staged_products + k_staged_rows (two streaming phases, x and y in LDS)"}


def fused_spmv_bytes(kernels, fmt, post_scaled, windows=True):
    """The fused SpMV instantiation of a PMC summary's kernel table (tools/pmc_summarize.py): 0 int32 indices, 1 offset codes,
    2 diagonal storage; post_scaled: the one-sweep steps' instantiation k_spmv_dia_win<true, NG, NCW, true>; windows: format 0 runs
    k_spmv_csr_win<EPI, ITERS, XI, PF, NT> (x windows in LDS, the default since round 5) rather than the gather kernel."""
    if fmt == 0 and windows:
        for name, rec in kernels.items():
            args_ = [a.strip() for a in name.split("<", 1)[1].rstrip(">").split(",")] if "<" in name else []
            if name.startswith("k_spmv_csr_win<true") and len(args_) == 5 and args_[4] == "false":
                return float(rec["hbm_bytes"])
        return None
    for name, rec in kernels.items():
        args_ = [a.strip() for a in name.split("<", 1)[1].rstrip(">").split(",")] if "<" in name else []
        if fmt == 2 and name.startswith("k_spmv_dia") and args_ and args_[0] == "true":
            # round 3 names: <EPI, NG, NCW, POST>; earlier rounds: <EPI, NG, FUSE, NCW> (no post-scaled variant)
            is_post = len(args_) == 4 and args_[3] == "true"
            if is_post == bool(post_scaled):
                return float(rec["hbm_bytes"])
        is_coded = len(args_) >= 4 and args_[3] == "true"
        if fmt != 2 and name.startswith("k_spmv_csr_stream<true") and is_coded == (fmt == 1):
            return float(rec["hbm_bytes"])
    return None


def pmc_traffic(n, fmt, post_scaled=False):
    """HBM bytes per launch of the fused SpMV as measured by the COMMITTED PMC passes (profiles/*pmc_traffic*.json), or None."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            if int(d.get("n", -1)) != int(n):
                continue
            b = fused_spmv_bytes(d["kernels"], fmt, post_scaled)
            if b is not None:
                return b, os.path.basename(path)
        except Exception:  # noqa: BLE001 - a malformed summary just means "no PMC figure"
            continue
    return None, None


def live_pmc_traffic(n, fmt, post_scaled, timeout=240, modes=None):
    """HBM bytes per launch of the fused SpMV measured IN THIS RUN: two rocprofv3 passes (FETCH_SIZE, then WRITE_SIZE — separate
    runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes) of tools/pmc_probe.py (the same matrix, the same
    instantiations, one factorisation sweep + restart per flow) in child processes, FETCH_SIZE calibrated on the probe's k_scale
    launches (8n bytes each way).  Returns (bytes or None, note).  Any failure — no rocprofv3, a pass that does not finish in
    `timeout` seconds (its process group is killed), no calibration kernel — leaves the committed figure in place."""
    import glob
    import importlib.util
    import shutil
    import signal
    import subprocess
    import tempfile

    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    # this process may itself run under a profiler (rocprofv3 -- python bench.py): no profiler inside a profiler
    under = [k for k in os.environ if k in ("ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_FORCE_LOAD") or k.startswith("ROCPROF_")]
    if under or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself profiled (" + ", ".join(under or ["LD_PRELOAD"]) + "): the live passes are skipped"
    tmp = tempfile.mkdtemp(prefix="mispec_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", PROBE_N=str(n), PROBE_FORMATS=str(fmt), PROBE_SPMV_REPS="0")
    if modes:
        env["PROBE_MODES"] = modes
    files = {}
    t0 = time.time()
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, counter), "-o", "p", "--",
                   sys.executable, os.path.join(ROOT, "tools", "pmc_probe.py")]
            child = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                child.wait(timeout=timeout)
            except subprocess.TimeoutExpired:
                os.killpg(child.pid, signal.SIGKILL)  # the session this call started, nothing else
                child.wait()
                return None, f"the {counter} pass did not finish in {timeout} s"
            found = glob.glob(os.path.join(tmp, counter, "**", "*counter_collection.csv"), recursive=True)
            if child.returncode != 0 or not found:
                return None, f"the {counter} pass failed (exit code {child.returncode})"
            files[counter] = found[0]
        spec = importlib.util.spec_from_file_location("pmc_summarize", os.path.join(ROOT, "tools", "pmc_summarize.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        d = mod.summarize(files["FETCH_SIZE"], files["WRITE_SIZE"], n)
        if not d["calibration"]["found"]:
            return None, "no k_scale launch in the probe to calibrate FETCH_SIZE on"
        b = fused_spmv_bytes(d["kernels"], fmt, post_scaled)
        if b is None:
            return None, "the probe did not launch the instantiation the solve used"
        return b, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes, --kernel-trace only) of "
                   f"tools/pmc_probe.py, {time.time() - t0:.0f} s; FETCH_SIZE x {d['calibration']['read']:.3f} (calibrated on k_scale: 8n bytes "
                   f"each way), WRITE_SIZE x {d['calibration']['write']:.3f}")
    except Exception as e:  # noqa: BLE001 - informational: the committed figure stays
        return None, f"live PMC passes failed: {e!r}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def flatten_for_the_driver(out):
    """The driver's record of a run keeps the SCALARS of the `roofline` block and drops nested objects and unknown top-level keys
    (BENCH_rNN.json `parsed`): the figures a reader of that record needs — the int32 CSR kernel on the headline matrix, the
    secondary workloads, the orthogonalisation pass, synchronisations per solve — are copied there as flat scalars (VERDICT r04
    item 1).  csr_kernel_frac is on SURVEY.md 8d's bytes: (12 nnz + 20 n + 4) / time / 8 TB/s."""
    sec = out["secondary"] if isinstance(out.get("secondary"), dict) else {}
    flat = out["roofline"]
    c32 = sec.get("csr_kernels_same_matrix", {}).get("csr_int32")
    if c32:
        flat["csr_kernel_ms"] = c32.get("ms_per_launch")
        flat["csr_kernel_frac"] = c32.get("frac_8d")
        flat["csr_kernel_frac_with_epilogue_operands"] = c32.get("frac")
        flat["csr_kernel_eigenpairs_per_s"] = c32.get("eigenpairs_per_s")
    for key in ("m_rand", "jitter_band", "stencil_rcm"):
        blk = sec.get(key, {})
        blk = blk.get("in_loop", blk) if isinstance(blk, dict) else {}
        if isinstance(blk, dict) and "frac" in blk:
            flat[f"secondary_{key}_frac"] = blk["frac"]
            flat[f"secondary_{key}_ms"] = blk["ms_per_launch"]
            flat[f"secondary_{key}_format"] = sec[key].get("spmv_format")
    if isinstance(sec.get("c4"), dict) and "seconds" in sec["c4"]:
        flat["secondary_c4_seconds"] = sec["c4"]["seconds"]
    if isinstance(sec.get("c5"), dict) and "seconds" in sec["c5"]:
        flat["secondary_c5_seconds"] = sec["c5"]["seconds"]
        flat["secondary_c5_solve_ms"] = sec["c5"].get("solve_ms")
    if out.get("roofline_orth"):
        flat["orth_frac"] = out["roofline_orth"].get("frac")
        if isinstance(out["roofline_orth"].get("pass_only"), dict):
            flat["orth_pass_frac"] = out["roofline_orth"]["pass_only"].get("frac")
    if isinstance(out.get("solve"), dict) and "host_syncs_per_solve" in out["solve"]:
        flat["host_syncs_per_solve"] = out["solve"]["host_syncs_per_solve"]
    if isinstance(out.get("solve"), dict) and out["solve"].get("host_turn_us") is not None:
        flat["host_turn_us"] = out["solve"]["host_turn_us"]
    sp_ = out.get("shard_proxy")
    if isinstance(sp_, dict) and "us_per_operation" in sp_:
        for k in ("rows", "us_per_operation", "idle_frac_est", "reduce_us", "host_turn_us", "host_syncs_per_solve", "speedup_8_compute_only"):
            flat[f"shard_proxy_{k}"] = sp_.get(k)
    # the three figures a reader of the headline asks for next, flat and at the top level too
    for name, val in (("value_reference_api", (out.get("value_with_host_eigenvectors") or {}).get("value")),
                      ("value_reference_flow", (out.get("other_orth_mode") or {}).get("value")
                       if (out.get("other_orth_mode") or {}).get("orth") == "reference" else None),
                      ("value_csr_kernel", flat.get("csr_kernel_eigenpairs_per_s"))):
        out[name] = val
        flat[name] = val


def spmv_block(op, ms, launches, fused, epi_vectors=2.0):
    """Roofline figures of one SpMV instantiation on the bytes it has to move.  epi_vectors: operands of the fused Lanczos epilogue —
    2 in the reference flow and the two-reduction one-sweep steps (v_prev and v), 1 in the one-reduction steps (only the operand of
    <f~, A f~>: nothing is subtracted in the kernel), a mean over the launches of a solve.  The operand v IS the product's input
    vector (already counted as x), so only v_prev adds bytes: 8 rows x (epi_vectors - 1).  (Rounds 1-4 added 16 rows, counting v
    twice; the PMC traffic of the one-reduction product on diagonal storage is 1.383 GB against 1.36 + 0.007 GB counted this way,
    profiles/r09v_pmc_traffic.json.)"""
    fmt = op.spmv_format()
    moved = op.stored_bytes() + (8.0 * max(epi_vectors - 1.0, 0.0) * op.local_rows() if fused else 0.0)
    gbps = moved / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {"kernel": KERNEL_OF_FORMAT[fmt], "ms_per_launch": ms, "launches": int(launches), "bytes_per_launch": moved,
            "achieved": gbps, "frac": gbps / HBM_PEAK_GBPS,
            "csr_equivalent_gbps": op.algorithmic_bytes() / (ms * 1e-3) / 1e9 if ms > 0 else 0.0}


def standalone_ms(op, ncols, reps):
    import torch

    x = torch.rand(ncols, dtype=torch.float64, device="cuda") - 0.5
    y = torch.empty(op.local_rows() + 2, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    op.spmv_time(x.data_ptr(), y.data_ptr(), 5)
    return op.spmv_time(x.data_ptr(), y.data_ptr(), reps)


def m_rand_host(n, seed=20240607):
    """M-rand of SURVEY.md 8d: every row has 7 partners at uniformly random columns, symmetrised (degrees vary around 15)."""
    from spectra_amd import workloads

    return workloads.m_rand(n, seed)


def in_loop_block(sa, ctx, rop, nev, ncv, rule, tol, restarts=12):
    """The SpMV of `rop` inside a solver loop (a bounded number of restarts: the in-loop time needs no convergence): figures on
    SURVEY.md 8d's bytes (12 nnz + 4 (rows + 1) + 8 cols + 8 rows — `frac`) and with the fused epilogue's two vector reads."""
    e = sa.SymEigsSolver(rop, nev, ncv)
    e.profile(2)
    e.init()
    e.compute(rule, 2, tol)  # warm-up: buffers, code paths
    p0 = e.get_profile()
    ctx.sync()
    t0 = time.perf_counter()
    e.init()
    nconv = e.compute(rule, restarts, tol)
    ctx.sync()
    dt = time.perf_counter() - t0
    p1 = e.get_profile()
    n_spmv = p1["n_spmv"] - p0["n_spmv"]
    ms = (p1["ms_spmv"] - p0["ms_spmv"]) / max(n_spmv, 1)
    alg = rop.algorithmic_bytes()
    blk = {"kernel": KERNEL_OF_FORMAT[rop.spmv_format()], "spmv_format": rop.spmv_format(), "ms_per_launch": ms, "launches": int(n_spmv),
           "bytes_per_launch": alg, "achieved": alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
           "bytes_note": "algorithmic: 12 nnz + 4 (rows + 1) + 8 cols + 8 rows (SURVEY.md 8d)",
           "frac_with_epilogue_operands": (alg + 8.0 * (1.0 - (e.orth_info()["one_reduction_steps"] / max(e.orth_info()["lagged_steps"], 1)
                                                                if e.orth_info().get("one_reduction") else 0.0)) * rop.local_rows())
                                          / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if ms > 0 else 0.0,
           "solve": {"seconds": dt, "restarts": restarts, "nconv": int(nconv), "num_operations": int(e.num_operations())}}
    blk["frac"] = blk["achieved"] / HBM_PEAK_GBPS
    del e
    return blk


def shard_proxy(args, ctx, sa, c2_ms_per_op, parts=8):
    """What ONE rank of an 8-way row-sharded run computes, on this GPU: the same solve on an M-band matrix of n / 8 rows (the
    wire is not in it: a 1-GPU box cannot measure it).  Un-instrumented wall time per solve and per operator application, the
    device-busy share (sum of the kernel families' HIP-event times of one more, instrumented solve), the merged record reduction
    on its own, the host turn of a restart, and the compute-only speed-up 8 ranks would have if the wire were free:
    (ms per operation at n) / (ms per operation at n / 8)."""
    rows = args.n // parts
    sop = sa.SparseSymMatProd.synth_band(rows, ctx=ctx)
    e = sa.SymEigsSolver(sop, args.nev, args.ncv)
    e.set_orth_mode(args.orth)
    rule = sa.SortRule[args.selection]

    def solve():
        e.init()
        nc = e.compute(rule, 1000, args.tol)
        e.eigenvectors(to_host=False)
        return nc

    solve()
    t_before = e.turn_info()
    p_before = e.get_profile()
    ctx.sync()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        nconv = solve()
    ctx.sync()
    wall_ms = 1e3 * (time.perf_counter() - t0) / reps
    t_after = e.turn_info()
    p_after = e.get_profile()
    nops = int(e.num_operations())
    e.profile(1)
    p0 = e.get_profile()
    solve()
    p1 = e.get_profile()
    e.profile(0)
    fam = {k[3:]: p1[k] - p0[k] for k in p1 if k.startswith("ms_")}
    busy_ms = sum(v for k, v in fam.items() if k in ("spmv", "vtf", "gemv", "scale", "compress", "small"))
    n_red = p1["n_reduce"] - p0["n_reduce"]
    turns = t_after["turns"] - t_before["turns"]
    out = {"rows": rows, "nconv": int(nconv), "num_operations": nops, "num_iterations": int(e.num_iterations()),
           "ms_per_solve": wall_ms, "us_per_operation": 1e3 * wall_ms / nops,
           "busy_ms_per_solve_instrumented": busy_ms,
           "idle_frac_est": ((1e3 * (t_after["host_seconds"] - t_before["host_seconds"]) / reps) / wall_ms) if turns else 0.0,
           "reduce_us": 1e3 * fam.get("reduce", 0.0) / n_red if n_red else None,
           "host_turn_us": 1e6 * (t_after["host_seconds"] - t_before["host_seconds"]) / turns if turns else None,
           "host_syncs_per_solve": (p_after["n_host_sync"] - p_before["n_host_sync"]) / reps,
           "speedup_8_compute_only": c2_ms_per_op / (wall_ms / nops) if c2_ms_per_op else None,
           "max_residual": float(e.residuals().max()),
           "note": "idle_frac_est = (host turns of a solve x their host time) / wall: the device has nothing to run while the host forms the "
                   "Ritz values and the restart's Q — every other gap of the trace is below a microsecond (rocprofv3 gap analysis of the same "
                   "workload: profiles/*trace_gaps_1250000_rows*); busy_ms_per_solve_instrumented brackets every family with events and is "
                   "inflated by them (it can exceed the wall time of the un-instrumented solves); "
                   "reduce_us is an event pair around the one-kernel record reduction (includes ~2 us of event overhead); host_turn_us = "
                   "host time between 'state of the finished sweep seen' and 'restart enqueued' (mispec_symeigs_turn_info)"}
    del e, sop
    return out


def secondary_configs(args, ctx, op, sa):
    """Driver-run figures of the other configurations (outside the timed region, one GPU)."""
    import numpy as np
    import scipy.sparse as sp

    out = {}
    rule = sa.SortRule[args.selection]

    # (1) the CSR kernels on the headline matrix, inside a solver loop (BASELINE.json words its roofline target for a CSR SpMV)
    csr = {}
    try:
        for fmt, name in ((0, "csr_int32"), (1, "csr_offset_codes")):
            op.set_spmv_format(fmt)
            if op.spmv_format() != fmt:
                continue
            e = sa.SymEigsSolver(op, args.nev, args.ncv)
            e.set_orth_mode(args.orth)
            e.profile(2)
            secs = None
            for rep in range(2):  # the second solve is the timed one (buffers allocated, code paths warm)
                p0 = e.get_profile()
                ctx.sync()
                t0 = time.perf_counter()
                e.init()
                nconv = e.compute(rule, 1000, args.tol)
                e.eigenvectors(to_host=False)
                ctx.sync()
                secs = time.perf_counter() - t0
            p1 = e.get_profile()
            n_spmv = p1["n_spmv"] - p0["n_spmv"]
            oi = e.orth_info()
            ev_mean = 2.0 - (oi["one_reduction_steps"] / max(oi["lagged_steps"], 1) if oi.get("one_reduction") else 0.0)
            blk = spmv_block(op, (p1["ms_spmv"] - p0["ms_spmv"]) / max(n_spmv, 1), n_spmv, True, ev_mean)
            blk["epilogue_vectors_mean"] = ev_mean
            blk["traffic"], blk["traffic_source"] = pmc_traffic(args.n, fmt)
            blk.update({"nconv": int(nconv), "num_operations": int(e.num_operations()), "seconds": secs,
                        "eigenpairs_per_s": nconv / secs, "orth": args.orth})
            blk["frac_8d"] = blk["csr_equivalent_gbps"] / HBM_PEAK_GBPS  # on SURVEY.md 8d's bytes (no epilogue operands)
            if fmt == 0:
                blk["windows"] = op.windows_info()
            csr[name] = blk
            del e
    finally:
        op.set_spmv_format(-1)
    out["csr_kernels_same_matrix"] = csr

    # (1b) ingest of the headline matrix from HOST arrays (the reference's operators take a host matrix and have no ingest cost):
    # the device-generated matrix is downloaded, then uploaded again as a full CSR matrix and as a lower triangle (CSC, the
    # reference's SparseSymMatProd<double, Lower> input).  Timed: the library call only (scipy conversions are done before).
    try:
        rp, ci, v = op.to_host_csr()
        full = sp.csr_matrix((v, ci, rp), shape=(args.n, args.n))
        t0 = time.perf_counter()
        up = sa.SparseGenMatProd(full, ctx=ctx)
        t_full = time.perf_counter() - t0
        st_full = sa.last_ingest_info()
        fmt_full = up.spmv_format()
        del up
        tri = sp.tril(full).tocsc()
        del full
        t0 = time.perf_counter()
        up = sa.SparseSymMatProd(tri, ctx=ctx)
        t_tri = time.perf_counter() - t0
        st_tri = sa.last_ingest_info()
        out["ingest_headline_matrix_from_host"] = {
            "full_csr": {"seconds": t_full, "stages": st_full, "spmv_format": fmt_full},
            "lower_triangle_csc": {"seconds": t_tri, "stages": st_tri, "spmv_format": up.spmv_format()},
            "host_threads": int(sa.lib().mispec_ingest_threads()), "nnz": int(len(v)),
            "note": "seconds = wall clock of the SparseGenMatProd / SparseSymMatProd constructor (Python marshalling + mispec_csr_upload / "
                    "mispec_csr_from_triangle); stages = the library's own breakdown (include/mispec.h mispec_last_ingest_info)"}
        del up, tri, rp, ci, v
    except Exception as e:  # noqa: BLE001
        out["ingest_headline_matrix_from_host"] = {"error": repr(e)}

    # (2) M-rand at the headline size: scattered columns, the gather-bound case
    t0 = time.perf_counter()
    A = m_rand_host(args.n)
    t_gen = time.perf_counter() - t0
    tri = sp.tril(A).tocsc()
    t0 = time.perf_counter()
    rop = sa.SparseSymMatProd(tri, ctx=ctx)
    t_ingest = time.perf_counter() - t0
    ingest_stages = sa.last_ingest_info()
    del tri
    alone = standalone_ms(rop, args.n, 20)
    e = sa.SymEigsSolver(rop, args.nev, args.ncv)
    e.profile(2)
    e.init()
    t0 = time.perf_counter()
    nconv = e.compute(rule, 12, args.tol)  # a bounded number of restarts: the in-loop SpMV time does not need convergence
    ctx.sync()
    dt = time.perf_counter() - t0
    p = e.get_profile()
    oi = e.orth_info()
    ev_mean = 2.0 - (oi["one_reduction_steps"] / max(oi["lagged_steps"], 1) if oi.get("one_reduction") else 0.0)
    inloop = spmv_block(rop, p["ms_spmv"] / max(p["n_spmv"], 1), p["n_spmv"], True, ev_mean)
    rop.set_spmv_format(0)
    csr_alone = spmv_block(rop, standalone_ms(rop, args.n, 10), 10, False)
    rop.set_spmv_format(-1)
    # M-rand's figures are quoted on the ALGORITHMIC bytes of SURVEY.md 8d (CSR with int32 indices, x counted once, + the fused
    # epilogue's two vector reads in the loop) whatever format ran, so that they stay comparable across rounds and formats: the
    # staged format (4) moves about 2.2 x those bytes by design — its rate on its own traffic is reported next to it.
    standalone = spmv_block(rop, alone, 20, False)
    for blk, extra in ((inloop, 8.0 * max(ev_mean - 1.0, 0.0) * rop.local_rows()), (standalone, 0.0)):
        alg = rop.algorithmic_bytes() + extra
        ms = blk["ms_per_launch"]
        blk.update({"moved_bytes_per_launch": blk["bytes_per_launch"], "achieved_on_moved_bytes": blk["achieved"], "frac_on_moved_bytes": blk["frac"],
                    "bytes_per_launch": alg, "achieved": alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0})
        blk["frac"] = blk["achieved"] / HBM_PEAK_GBPS
        blk["bytes_note"] = "algorithmic: 12 nnz + 4 (rows + 1) + 8 cols + 8 rows" + (f" + 8 rows x {max(ev_mean - 1.0, 0.0):.2f} (v_prev of the two-reduction steps)" if extra else "")
    out["m_rand"] = {"n": args.n, "nnz": rop.nnz(), "spmv_format": rop.spmv_format(), "reordering": rop.reordering(), "tiles": rop.tiles_info(),
                     "staged": rop.staged_info(),
                     "standalone": standalone, "in_loop": inloop, "standalone_csr_int32_kernel": csr_alone,
                     "ingest_seconds": t_ingest, "ingest_stages": ingest_stages, "ingest_host_threads": int(sa.lib().mispec_ingest_threads()),
                     "solve_12_restarts": {"seconds": dt, "nconv": int(nconv), "num_operations": int(e.num_operations())},
                     "host_generation_seconds": t_gen}
    del e, rop, A

    # (2b) irregular-but-local patterns at the headline size: neither M-band's fixed diagonals nor M-rand's far
    # gathers.  jitter band: M-band's off-diagonals moved by a per-entry jitter of <= 64 columns (variable row lengths, ~1800
    # distinct diagonals: no offset codes, no diagonal storage — int32 CSR is the automatic choice); 7-point stencil on a 215^3 grid
    # handed over in random order and reordered at ingest (reverse Cuthill-McKee).  Both run the int32 CSR kernel with x windows.
    from spectra_amd import workloads
    for key, make in (("jitter_band", lambda: workloads.jitter_band(args.n)), ("stencil_rcm", None)):
        try:
            t0 = time.perf_counter()
            if make is not None:
                A = make()
            else:
                B = workloads.stencil7(max(8, int(round(args.n ** (1.0 / 3.0)))))
                perm = np.random.default_rng(1).permutation(B.shape[0])
                A = B[perm][:, perm].tocsr()
                A.sort_indices()
                del B, perm
            t_gen = time.perf_counter() - t0
            tri = sp.tril(A).tocsc()
            nloc = A.shape[0]
            del A
            t0 = time.perf_counter()
            rop = sa.SparseSymMatProd(tri, ctx=ctx)
            t_ingest = time.perf_counter() - t0
            del tri
            blk = in_loop_block(sa, ctx, rop, args.nev, args.ncv, rule, args.tol)
            blk.update({"n": nloc, "nnz": rop.nnz(), "reordering": rop.reordering(), "windows": rop.windows_info(),
                        "standalone_ms_per_launch": standalone_ms(rop, nloc, 20), "host_generation_seconds": t_gen, "ingest_seconds": t_ingest})
            auto_windows = rop.nnz() >= 9 * nloc and blk["windows"]["lds_doubles"] > 0  # csr.hpp windows_active()
            blk["kernel_in_use"] = "k_spmv_csr_win (x windows in LDS)" if auto_windows else "k_spmv_csr_stream (gathers: fewer than 9 entries per row)"
            rop.use_windows(not auto_windows)
            blk["other_kernel_in_loop_ms"] = in_loop_block(sa, ctx, rop, args.nev, args.ncv, rule, args.tol, 6)["ms_per_launch"]
            rop.use_windows(None)
            out[key] = blk
            del rop
        except Exception as e:  # noqa: BLE001
            out[key] = {"error": repr(e)}

    # (3) C4: GenEigsSolver on the 5M non-symmetric band matrix, k = 10, ncv = 30
    gop = sa.SparseGenMatProd.synth_band(5_000_000, ctx=ctx)
    best = None
    for r in range(3):
        g = sa.GenEigsSolver(gop, 10, 30)
        g.profile(2)
        ctx.sync()
        t0 = time.perf_counter()
        g.init()
        nconv = g.compute(sa.SortRule.LargestMagn, 1000, 1e-11)
        ctx.sync()
        dt = time.perf_counter() - t0
        if r > 0 and (best is None or dt < best[0]):
            best = (dt, g, nconv)
    dt, g, nconv = best
    p = g.get_profile()
    out["c4"] = {"config": "GenEigsSolver 5M x 5M non-symmetric CSR, k=10, ncv=30, LargestMagn, tol 1e-11", "seconds": dt,
                 "eigenpairs_per_s": nconv / dt, "nconv": int(nconv), "num_operations": int(g.num_operations()),
                 "max_residual": float(g.residuals().max()), "spmv": spmv_block(gop, p["ms_spmv"] / max(p["n_spmv"], 1), p["n_spmv"], False)}
    del g, gop, best

    # (4) C5: shift-and-invert on the 2M banded matrix, sigma = 0, k = 6, ncv = 20
    n5, b = 2_000_000, 3
    rng = np.random.default_rng(5)
    diags = [rng.uniform(-0.5, 0.5, n5 - d) for d in range(1, b + 1)]
    A5 = sp.diags([rng.uniform(-0.5, 0.5, n5) + b + 0.5] + diags + diags, [0] + list(range(1, b + 1)) + [-d for d in range(1, b + 1)], format="csc")
    sop = sa.SparseSymShiftSolve(sp.tril(A5).tocsc(), ctx=ctx)
    sop.set_shift(0.0)
    t0 = time.perf_counter()
    sop.set_shift(0.0)
    t_factor = time.perf_counter() - t0
    best = None
    for r in range(3):
        s = sa.SymEigsShiftSolver(sop, 6, 20, 0.0)
        s.profile(2)
        ctx.sync()
        t0 = time.perf_counter()
        s.init()
        nconv = s.compute(sa.SortRule.LargestMagn, 1000, 1e-11)
        ctx.sync()
        dt = time.perf_counter() - t0
        if r > 0 and (best is None or dt < best[0]):
            best = (dt, s, nconv)
    dt, s, nconv = best
    p = s.get_profile()
    solve_ms = p["ms_spmv"] / max(p["n_spmv"], 1)
    # Bytes of one solve.  `solve_bytes`: the lower bound — every factor entry, spike and vector touched once:
    # (b + 1) factor + 2b spikes + rhs + solution.  `solve_bytes_moved`: what the kernels read and write — the factor is read by
    # both sweeps, f / y / x pass through three kernels, plus the second level's block inverses and the dense last level.
    solve_bytes = (2 * b + 1 + 2 * b) * 8.0 * n5
    n1 = n5 // 128 * b                              # rows of the second level, half-bandwidth 2b - 1
    blocks = (n1 // 128) * 128 * 128 * 8.0          # explicit inverses of its chunk interiors
    n2 = n1 // 128 * (2 * b - 1)                    # rows of the dense last level
    solve_bytes_moved = ((2 * b + 1) + 2 + (2 * b + 2)) * 8.0 * n5 + blocks + (2 * (2 * b - 1) + 4) * 8.0 * n1 + 8.0 * n2 * n2
    out["c5"] = {"config": "SymEigsShiftSolver 2M x 2M banded (half-bandwidth 3), sigma=0, k=6, ncv=20, tol 1e-11", "seconds": dt,
                 "eigenpairs_per_s": nconv / dt, "nconv": int(nconv), "num_operations": int(s.num_operations()),
                 "set_shift_seconds": t_factor, "solve_ms": solve_ms, "solve_bytes": solve_bytes,
                 "solve_frac_of_hbm_peak": solve_bytes / (solve_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if solve_ms > 0 else None,
                 "solve_bytes_moved": solve_bytes_moved,
                 "solve_moved_frac_of_hbm_peak": solve_bytes_moved / (solve_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if solve_ms > 0 else None}
    return out


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(respawn_as_ranks(args))
    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner when a communicator
    # is created): everything this process and its libraries print goes to stderr, the line itself to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    import torch.distributed as dist

    import spectra_amd as sa
    from spectra_amd import dist as sdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rccl_log = None
    if world > 1 and os.environ.get("MISPEC_COMM") != "gloo-staged" and "NCCL_DEBUG" not in os.environ:
        # the first multi-GPU run has to be diagnostic without a second try: RCCL's own account of the
        # topology and of the algorithm / protocol / channels it picked goes to a per-rank file, rank 0's digest into the line
        rccl_log = f"/tmp/mispec_rccl_rank{os.environ.get('RANK', '0')}_{os.getpid()}.log"
        os.environ["NCCL_DEBUG"] = "INFO"
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH,ENV")
        os.environ["NCCL_DEBUG_FILE"] = rccl_log
    if world > 1 or args.gpus > 1 or os.environ.get("MISPEC_FORCE_COMM") == "1":
        # MISPEC_COMM=gloo-staged (tests on a box with fewer GPUs than ranks): gloo process group, collectives staged through
        # host memory, ranks beyond the device count share the last device — exercises this file's N > 1 path end to end
        staged = os.environ.get("MISPEC_COMM") == "gloo-staged"
        rank, world = sdist.init_process_group("gloo" if staged else "nccl")
        assert world == args.gpus, f"launched {world} ranks for --gpus {args.gpus}"
        using_dist = True
    else:
        rank, world = 0, 1
        using_dist = False
    local = min(int(os.environ.get("LOCAL_RANK", "0")), torch.cuda.device_count() - 1)
    torch.cuda.set_device(local)
    ctx = sdist.make_context(local)

    def barrier():
        torch.cuda.synchronize()
        ctx.sync()
        if world > 1:
            dist.barrier()

    op = sa.SparseSymMatProd.synth_band(args.n, ctx=ctx)  # resident in HBM before any timing
    nnz_local = op.nnz()
    rule = sa.SortRule[args.selection]

    def new_solver(profile, orth=None):
        """One solver object = one set of device buffers (V 3.2 GB, X 1.6 GB at n = 1e7): created OUTSIDE the timed region and
        re-used by every step, as a caller of the reference would re-use its SymEigsSolver (init() starts a new solve)."""
        eigs = sa.SymEigsSolver(op, args.nev, args.ncv)
        eigs.set_orth_mode(orth or args.orth)
        if profile:
            eigs.profile(profile)
        return eigs

    def solve(eigs):
        eigs.init()
        nconv = eigs.compute(rule, 1000, args.tol)
        ncols = eigs.eigenvectors(to_host=False)  # V * Y formed in HBM (1.6 GB at n = 1e7; not pulled over PCIe)
        return nconv, ncols

    def timed_steps(eigs, steps):
        """`steps` solves between barriers; returns (max-over-ranks seconds, pairs, profile of exactly these steps)."""
        before = eigs.get_profile()
        barrier()
        t0 = time.perf_counter()
        pairs = 0
        for _ in range(steps):
            nconv, _ = solve(eigs)
            pairs += nconv
        barrier()
        dt = sdist.max_over_ranks(time.perf_counter() - t0)
        after = eigs.get_profile()
        return dt, pairs, {k: (after[k] - before[k] if k != "spmv_bytes" else after[k]) for k in after}

    exchange_note = None
    # level 2: HIP events bracket only the operator applications (the roofline figure is measured live in the timed region); the
    # other families would cost ~10 more event records per Lanczos step
    eigs = new_solver(0 if args.no_profile else args.profile_level)
    if world > 1:
        # Self-check of the point-to-point neighbour exchange (outside the timed region): if the solve it drives
        # does not reach the residual bar on every rank, every rank falls back to the plain all-gather.
        nconv_chk, _ = solve(eigs)
        r = eigs.residuals()
        bad = int(nconv_chk < args.nev or not np.all(np.isfinite(r)) or float(r.max()) > 1e-8)
        flag = torch.tensor([bad], dtype=torch.int32, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) and eigs.exchange_info()[0]:
            os.environ["MISPEC_EXCHANGE"] = "allgather"
            exchange_note = "neighbour exchange failed its residual self-check; all-gather used"
            del eigs
            eigs = new_solver(0 if args.no_profile else args.profile_level)  # the exchange is planned at construction
    for _ in range(args.warmup):
        solve(eigs)
    elapsed, total_pairs, prof = timed_steps(eigs, args.steps)

    # ---- everything below is outside the timed region -------------------------------------------------
    halo, recv_doubles = eigs.exchange_info()
    # north_star words the exchange as an all-gather of the Krylov vector: when the timed region used the neighbour
    # exchange, the same number of steps is also run with the all-gather and reported next to it
    allgather_run = None
    if world > 1 and halo:
        prev = os.environ.get("MISPEC_EXCHANGE")
        os.environ["MISPEC_EXCHANGE"] = "allgather"
        ag = new_solver(2)
        solve(ag)
        ag_elapsed, ag_pairs, ag_prof = timed_steps(ag, args.steps)
        allgather_run = {"value": ag_pairs / ag_elapsed, "ms_per_step": 1e3 * ag_elapsed / args.steps,
                         "spmv_ms_per_launch_incl_exchange_wait": ag_prof["ms_spmv"] / max(ag_prof["n_spmv"], 1),
                         "moved_mb_per_product_per_rank": (world - 1) * int(sa.lib().mispec_shard_block(args.n, world)) * 8 / 1e6}
        del ag
        if prev is None:
            del os.environ["MISPEC_EXCHANGE"]
        else:
            os.environ["MISPEC_EXCHANGE"] = prev
    # the wire, timed on its own (profile level 3: HIP events around the exchange on its stream, around the part of it a product
    # waits for, around the all-reduces) in one more solve outside the timed region — flat `wire_*` scalars in the line
    wire = None
    if world > 1:
        we = new_solver(3)
        solve(we)
        w0 = we.get_profile()
        barrier()
        t0 = time.perf_counter()
        solve(we)
        barrier()
        w_dt = sdist.max_over_ranks(time.perf_counter() - t0)
        w1 = we.get_profile()
        wd = {k: w1[k] - w0[k] for k in w1 if k != "spmv_bytes"}
        nops_w = max(wd["n_spmv"], 1)
        wire = {"ms_per_solve_instrumented": 1e3 * w_dt, "operations": int(wd["n_spmv"]),
                "exchange_us": 1e3 * wd["ms_exchange"] / max(wd["n_exchange"], 1), "exchanges": int(wd["n_exchange"]),
                "exchange_wait_us": 1e3 * wd["ms_exchange_wait"] / max(wd["n_exchange_wait"], 1), "exchange_waits": int(wd["n_exchange_wait"]),
                "overlap_frac": (1.0 - wd["ms_exchange_wait"] / wd["ms_exchange"]) if wd["ms_exchange"] > 0 else None,
                "allreduce_us": 1e3 * wd["ms_allreduce"] / max(wd["n_allreduce"], 1), "allreduces": int(wd["n_allreduce"]),
                "allreduces_per_operation": wd["n_allreduce"] / nops_w,
                "wire_ms_per_solve": wd["ms_exchange_wait"] + wd["ms_allreduce"],
                "spmv_ms_per_launch_incl_exchange_wait": wd["ms_spmv"] / nops_w,
                "note": "rank 0's HIP events (mispec_profile level 3) of ONE extra solve: exchange = the neighbour exchange / all-gather of the Krylov "
                        "vector on its own stream; exchange_wait = what the product still waits for once the interior row blocks are done "
                        "(overlap_frac = 1 - wait / exchange); allreduce = the record / alpha all-reduces on the compute stream; "
                        "wire_ms_per_solve = exchange_wait + allreduce, the wire's share of the critical path"}
        del we
    rccl_info = None
    if rccl_log and rank == 0:
        try:
            keys = ("Ring", "Tree", "Channel 00", "nChannels", "comm 0x", "via ", "XGMI", "P2P", "nNodes", "Algo", "Proto", "NCCL_", "RCCL", "HSA_", "topo")
            with open(rccl_log) as f:
                hits = [ln.strip()[-220:] for ln in f if any(k in ln for k in keys)]
            rccl_info = hits[:40]
            sys.stderr.write("[bench] RCCL (NCCL_DEBUG=INFO, rank 0, first matching lines):\n" + "\n".join(rccl_info[:40]) + "\n")
        except OSError as ex:
            rccl_info = [repr(ex)]
    # the other orthogonalisation mode on the same matrix, same number of steps (not part of `value`)
    other_mode = None
    if True:  # always reported, with or without profiling
        other = "reference" if args.orth.startswith("onesweep") else "onesweep"
        oe = new_solver(0, other)
        solve(oe)
        o_elapsed, o_pairs, _ = timed_steps(oe, args.steps)
        o_res = oe.residuals()
        other_mode = {"orth": other, "value": o_pairs / o_elapsed, "ms_per_step": 1e3 * o_elapsed / args.steps,
                      "num_operations": int(oe.num_operations()), "num_iterations": int(oe.num_iterations()),
                      "max_residual": float(o_res.max()) if len(o_res) else None, "orth_info": oe.orth_info()}
        del oe
    # the reference's eigenvectors() hands back a HOST matrix: the same steps with that copy inside the timed loop
    with_host = None
    if world == 1 and not args.no_profile:
        barrier()
        t0 = time.perf_counter()
        hp = 0
        for _ in range(args.steps):
            eigs.init()
            hp += eigs.compute(rule, 1000, args.tol)
            Xh = eigs.eigenvectors()
        barrier()
        with_host = {"value": hp / (time.perf_counter() - t0), "note": f"eigenvectors() returned as a {Xh.shape[0]} x {Xh.shape[1]} host matrix "
                     "(a FRESH pageable numpy array per solve, D2H inside the timed loop) as the reference's API does"}
        # ... and into the SAME host matrix every time (eigenvectors(out=...), C++: eigenvectors_to): without the first-touch page faults
        barrier()
        t0 = time.perf_counter()
        hp = 0
        for _ in range(args.steps):
            eigs.init()
            hp += eigs.compute(rule, 1000, args.tol)
            eigs.eigenvectors(out=Xh)
        barrier()
        with_host["value_into_a_reused_host_matrix"] = hp / (time.perf_counter() - t0)
        del Xh
    # per-family kernel split: one more solve with every family instrumented, not part of `value`
    split = None
    pass_only = None
    if not args.no_profile:
        full = new_solver(1)
        solve(full)
        split = full.get_profile()
        del full
        # ... and one with ONLY the passes over the basis bracketed (level 4): with every family bracketed the event pairs
        # themselves stretch each 0.5 ms pass by ~10 % (they keep the next launch from being prepared behind the running
        # kernel); alone they cost ~1 % — the figure that agrees with the rocprofv3 trace of the same kernels
        po = new_solver(4)
        solve(po)
        pass_only = po.get_profile()
        del po
    resid = eigs.residuals()
    evals = eigs.eigenvalues()
    turn1 = eigs.turn_info()
    spmv_ms = prof["ms_spmv"] / max(prof["n_spmv"], 1)
    fmt = op.spmv_format()
    oinfo = eigs.orth_info()
    onered_frac = oinfo["one_reduction_steps"] / max(oinfo["lagged_steps"], 1) if oinfo.get("one_reduction") else 0.0
    epi_vectors = 2.0 - onered_frac  # one-reduction steps read one epilogue operand (f~ for <f~, A f~>), the others v_prev and v
    head = spmv_block(op, spmv_ms, prof["n_spmv"], True, epi_vectors)
    alone_ms = standalone_ms(op, args.n if world == 1 else int(sa.lib().mispec_shard_block(args.n, world)) * world, args.spmv_reps)

    transport_name = {"gloo-staged": "gloo (host-staged)", "torch": "torch.distributed (RCCL)"}.get(os.environ.get("MISPEC_COMM", "rccl"), "RCCL")
    if type(getattr(ctx, "_comm", None)).__name__ == "TorchComm":  # (also the fallback when the library's own communicator failed)
        transport_name = "torch.distributed (RCCL)"
    if world == 1:
        exchange_desc = ""
    elif halo:
        exchange_desc = (f", {transport_name} point-to-point exchange of the referenced parts of the Krylov vector per SpMV "
                         f"({recv_doubles * 8 / 1e6:.2f} MB received by rank 0; the all-gather would move "
                         f"{(world - 1) * int(sa.lib().mispec_shard_block(args.n, world)) * 8 / 1e6:.1f} MB; "
                         f"the all-gather variant is timed as well: `allgather_variant`)")
    else:
        exchange_desc = f", {transport_name} all-gather of the Krylov vector per SpMV" + (f" ({exchange_note})" if exchange_note else "")
    if rank == 0:
        # the instantiation most launches of a solve use: one-reduction steps run the plain fused kernel (no post-scaling), the
        # two-reduction one-sweep steps on diagonal storage the post-scaled one
        post_inst = args.orth.startswith("onesweep") and onered_frac < 0.5
        traffic, traffic_file = pmc_traffic(args.n, fmt, post_scaled=post_inst) if world == 1 else (None, None)
        if args.orth.startswith("onesweep") and not post_inst and traffic_file and not traffic_file.startswith("r09"):
            traffic, traffic_file = None, None  # (summaries of earlier rounds mix this instantiation's reference-flow launches in)
        traffic_committed, live_note = traffic, "not attempted (--no-live-pmc or more than one rank)"
        if world == 1 and not args.no_live_pmc:
            live, live_note = live_pmc_traffic(args.n, fmt, post_inst, modes=args.orth.replace("-eager", ""))
            if live is not None:
                traffic, traffic_file = live, None
        out = {
            "metric": "eigenpairs_per_sec",
            "value": total_pairs / elapsed,
            "unit": "eigenpairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": ("BASELINE.json configs[1]: SymEigsSolver + SparseSymMatProd, 10M x 10M ~15 nnz/row fp64 "
                             "symmetric CSR (M-band, SURVEY.md 8d)" if args.n == 10_000_000 else f"M-band n={args.n}"),
                "n": args.n, "nnz_per_gpu": nnz_local, "nev": args.nev, "ncv": args.ncv, "selection": args.selection,
                "tol": args.tol, "start_vector": "SimpleRandom(0) (reference default)",
                "parallelism": f"row-shard x{world}" + exchange_desc,
                "orthogonalisation": ({"reference": "reference control flow (Lanczos.h:145-181): V'f, then f -= Vc with |f| and the V'f check — "
                                                    "two passes over V per step",
                                       "onesweep": "the library default: one-sweep steps (mispec_fac_set_orth_mode, DESIGN.md 3.2.1): the correction of a step "
                                                   "rides on the next step's pass over V; same decisions and fixed points, parity-gated by "
                                                   "tests/test_gpu_onesweep.py, the reference's own test programs in both modes and the full-size "
                                                   "golden; the reference-flow (MISPEC_ORTH=reference) figure of the same "
                                                   "run is `other_orth_mode`; the last correction of every sweep rides on the restart's "
                                                   "V*Q pass (k_vq_fused); since round 5 with ONE reduction per lagged step (the product runs on the "
                                                   "un-normalised residual, its <f~, A f~> is reduced with the record of the previous pass; "
                                                   "MISPEC_ONE_REDUCTION=0 restores two) and a restart that needs no host turn"}[args.orth.replace("-eager", "")]),
                "solver_object": "one SymEigsSolver (V, X, work vectors) allocated before the timed region and re-used by every step",
                "eigenvectors": ("X = V*Y is formed in HBM and left there (the reference's eigenvectors() returns a host matrix: "
                                 f"the D2H copy of {8e-9 * args.n * args.nev:.1f} GB would add ~{8e-9 * args.n * args.nev / 55 * 1e3:.0f} ms per solve "
                                 "at ~55 GB/s PCIe and is not part of `value`)"),
            },
            "roofline": {
                "kernel": head["kernel"] + (" (one-reduction steps: u = A f~ fused with the partial sums of <f~, u>; w = u/beta - beta*v_prev is formed by the "
                                            "orthogonalisation pass)" if onered_frac >= 0.5 else
                                            " (SpMV fused with w -= beta*v_prev and the alpha dot" +
                                            ("; one-sweep steps: input f, row sums divided by beta" if args.orth.startswith("onesweep") else "") + ")"),
                "bound": "hbm",
                "achieved": head["achieved"],
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": head["frac"],
                "traffic": traffic,
                "traffic_source": ((f"profiles/{traffic_file}: HBM bytes per launch of the in-loop SpMV instantiation from the committed rocprofv3 PMC "
                                    f"passes (FETCH_SIZE x calibration + WRITE_SIZE, separate runs; see profiles/README.md); live passes: {live_note}")
                                   if traffic_file or traffic is None else live_note),
                "traffic_committed": traffic_committed,
                "bytes_per_launch": head["bytes_per_launch"],
                "bytes_note": {0: "CSR int32: 12 nnz + 4 (rows+1) + 8 cols + 8 rows (SURVEY.md 8d)",
                               1: "offset-coded CSR: 9 nnz + 4 (rows+1) + 8 cols + 8 rows",
                               2: "diagonal storage: 8 ndia rows + 8 cols + 8 rows — the bytes this kernel has to move, not the CSR figure"}[fmt] +
                              f" + 8 rows x {max(epi_vectors - 1.0, 0.0):.2f} for v_prev (read by the two-reduction steps only; the epilogue's other "
                              "operand is the input vector itself, counted as x)",
                "epilogue_vectors_mean": epi_vectors,
                "index_format": {0: "CSR, int32 column indices",
                                 1: f"CSR, offset codes: 1 byte per entry into {op.offset_codes()} diagonals",
                                 2: f"diagonal storage: {op.offset_codes()} diagonals, no index, no gather"}[fmt],
                "ms_per_launch": spmv_ms,
                "launches": int(prof["n_spmv"]),
                "csr_equivalent_gbps": head["csr_equivalent_gbps"],
                "csr_equivalent_note": "SURVEY.md 8d's CSR/int32 byte count (12 nnz + 20 n) over the same time: comparable across formats and "
                                       "rounds, but not an HBM rate unless the int32 CSR kernel ran",
                "standalone_ms_per_launch": alone_ms,
                "standalone_gbps": op.stored_bytes() / (alone_ms * 1e-3) / 1e9,
            },
            "solve": {
                "nconv": int(total_pairs // args.steps), "num_operations": int(eigs.num_operations()),
                "num_iterations": int(eigs.num_iterations()), "max_residual": float(resid.max()) if len(resid) else None,
                "lambda_max": float(evals.max()) if len(evals) else None, "lambda_min": float(evals.min()) if len(evals) else None,
                "host_syncs_per_solve": prof["n_host_sync"] / args.steps,
                "host_turn_us": (1e6 * turn1["host_seconds"] / turn1["turns"]) if turn1["turns"] else None,
                "host_turn_copy_fallbacks": turn1["fallbacks"],
                "orth_info": eigs.orth_info(),
            },
            "other_orth_mode": other_mode,
            "value_with_host_eigenvectors": with_host,
            "roofline_orth": ({
                "kernel": {"onesweep": "k_orth_lagged: correction of step i-1 + projection of step i in one pass over V (the largest share of a solve)",
                           "reference": "k_orth<RESID_VTF>: f = w - alpha v, |f|, V'f (the first of the two passes over V of a step)"}[args.orth.replace("-eager", "")],
                "bound": "hbm", "bytes_per_launch_mean": split["bytes_vtf"] / max(split["n_vtf"], 1),
                "ms_per_launch_mean": split["ms_vtf"] / max(split["n_vtf"], 1),
                "achieved": split["bytes_vtf"] / (split["ms_vtf"] * 1e-3) / 1e9 if split["ms_vtf"] > 0 else None,
                "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": split["bytes_vtf"] / (split["ms_vtf"] * 1e-3) / 1e9 / HBM_PEAK_GBPS if split["ms_vtf"] > 0 else None,
                "note": "algorithmic bytes (8 n x vectors read + written, summed over the launches of one solve) over the family's HIP-event "
                        "time, which also covers the ~10 us record reduction behind every pass; from the instrumented extra solve",
                "pass_only": ({"ms_per_solve": pass_only["ms_vtf"], "bytes_per_solve": pass_only["bytes_vtf"],
                               "passes": int(pass_only["n_vtf"] - pass_only["n_reduce"]),
                               "ms_per_pass_mean": pass_only["ms_vtf"] / max(pass_only["n_vtf"] - pass_only["n_reduce"], 1),
                               "achieved": pass_only["bytes_vtf"] / (pass_only["ms_vtf"] * 1e-3) / 1e9,
                               "frac": pass_only["bytes_vtf"] / (pass_only["ms_vtf"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                               "note": "one more solve with ONLY the passes bracketed by HIP events (mispec_profile level 4, no record "
                                       "reductions inside): the kernels' own time, agreeing with rocprofv3's AverageNs of k_orth_lagged_dma "
                                       "(profiles/r11f_c2_only_dia_kernel_stats.csv); `frac` above brackets every family and is ~10 % "
                                       "pessimistic for it"} if pass_only and pass_only["ms_vtf"] > 0 else None),
                "compress": {"kernel": ("k_vq_fused (V <- V Q in place with the pending correction of the sweep's last step, its V'f test and the "
                                        "restarted residual on the same tiles), k_vq (X = V Y)" if args.orth == "onesweep"
                                        else "k_vq (V <- V Q in place, X = V Y)"), "achieved": split["bytes_compress"] / (split["ms_compress"] * 1e-3) / 1e9
                             if split["ms_compress"] > 0 else None}} if split else None),
            "kernels_ms_per_solve": ({k[3:]: split[k] for k in split if k.startswith("ms_")} if split else None),
            "kernels_ms_note": "from one additional solve with every kernel family bracketed by HIP events, outside the timed region",
            "kernels_launches_per_solve": {k[2:]: prof[k] / args.steps for k in prof if k.startswith("n_")},
        }
        if world > 1:
            block_mb = int(sa.lib().mispec_shard_block(args.n, world)) * 8 / 1e6
            out["collectives_per_step"] = {
                "exchange": {"count": 1, "kind": "neighbour send/recv" if halo else "all-gather",
                             "megabytes_received_per_rank": recv_doubles * 8 / 1e6 if halo else (world - 1) * block_mb},
                "all_reduce_sum": ([{"what": "ONE message per lagged step (include/mispec.h MISPEC_ORTH_ONE_REDUCTION): the record of the pass of step i "
                                             "— c' = [V, v_i]'f~ (i + 1 sums), chk = V'v_i (i), |f~|^2 — and, behind it, <f~, A f~> of the product of "
                                             "step i + 1, which ran on the un-normalised residual", "bytes_max": 8 * (2 * args.ncv + 1)}]
                                   if args.orth.startswith("onesweep") and eigs.orth_info().get("one_reduction") else
                                   [{"what": "alpha = <v, w>", "bytes": 8}] +
                                   ([{"what": "record of the one-sweep pass of step i: c' = [V, v_i]'f (i + 1 sums), chk = V'v_i (i), |f|^2 — one "
                                              "contiguous message", "bytes_max": 8 * 2 * args.ncv}] if args.orth.startswith("onesweep") else
                                    [{"what": "record: V'f (i + 1 sums), |f|^2 — one contiguous message", "bytes_max": 8 * (args.ncv + 1)}] * 2)),
                "note": "per Lanczos step and rank; one-sweep: one record per step (the first step of a sweep takes the two-reduction form; the "
                        "last correction of a sweep and its test ride on the restart's V*Q pass: one more record of ncv + 2 sums per restart), "
                        "reference flow: alpha, V'f and the correction's V'f check"}
        if allgather_run:
            out["allgather_variant"] = allgather_run
        if wire:
            out["wire"] = wire
            for k in ("exchange_us", "exchange_wait_us", "overlap_frac", "allreduce_us", "allreduces_per_operation", "wire_ms_per_solve"):
                out["wire_" + k if not k.startswith("wire_") else k] = wire[k]
        if rccl_info is not None:
            out["rccl_info"] = rccl_info
        if world == 1 and not args.no_secondary and not args.no_profile:
            try:
                out["secondary"] = secondary_configs(args, ctx, op, sa)
            except Exception as e:  # noqa: BLE001 - the headline line must still be printed
                out["secondary"] = {"error": repr(e)}
            try:
                out["shard_proxy"] = shard_proxy(args, ctx, sa, out["ms_per_step"] / max(out["solve"]["num_operations"], 1))
            except Exception as e:  # noqa: BLE001
                out["shard_proxy"] = {"error": repr(e)}
            # north_star names a CSR SpMV: the int32 CSR kernel's in-loop figure on the SAME matrix sits next to the headline
            # kernel's, so that a regression of either is visible in the top-level block
            c32 = out["secondary"].get("csr_kernels_same_matrix", {}).get("csr_int32") if isinstance(out["secondary"], dict) else None
            flatten_for_the_driver(out)
            if c32:
                out["roofline"]["csr_kernel"] = {
                    "kernel": c32.get("kernel"), "ms_per_launch": c32.get("ms_per_launch"), "bytes_per_launch": c32.get("bytes_per_launch"),
                    "achieved": c32.get("achieved"), "frac": c32.get("frac"), "csr_equivalent_gbps": c32.get("csr_equivalent_gbps"),
                    "traffic": c32.get("traffic"), "eigenpairs_per_s": c32.get("eigenpairs_per_s"),
                    "note": "the int32 CSR kernel (x windows in LDS) forced on the headline matrix (mispec_csr_set_spmv_format 0), "
                            "HIP events inside a complete solve; bytes = 12 nnz + 4 (rows+1) + 8 cols + 8 rows + 8 rows per v_prev read; "
                            "csr_kernel_frac above is on SURVEY.md 8d's bytes alone"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, int(eigs.num_operations()), int(total_pairs // args.steps), int(eigs.num_iterations()))
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if using_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
