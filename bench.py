#!/usr/bin/env python
"""Headline benchmark: BASELINE.json configs[1] — SymEigsSolver on a 10M x 10M, ~15 nnz/row fp64 symmetric CSR,
k = 20, ncv = 40 — on N MI355X of one node (N > 1: the same matrix row-partitioned, RCCL all-gather of the
Krylov vector per SpMV = configs[2], strong scaling).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

A "step" is ONE complete solve: init() + compute(LargestMagn, tol) + eigenvectors() with the matrix already
resident in HBM (generated there by the counter-hash generator of SURVEY.md §8d).  Rank 0 prints one JSON
line: value = eigenpairs/s of the whole job; `roofline` = the per-iteration CSR SpMV kernel, algorithmic
bytes / mean launch duration from HIP events recorded on the solver's stream inside the timed region;
`cpu_baseline` = the CPU oracle (Eigen-free restatement of the reference, 1 thread like the reference) timed on
a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


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
    p.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    p.add_argument("--profile-level", type=int, default=2, choices=[1, 2],
                    help="HIP-event instrumentation of the timed solves: 2 = operator applications only (default), 1 = every kernel family")
    p.add_argument("--cpu-steps", type=int, default=20, help="Lanczos steps of the CPU sample (about 10 s of one host core at n = 1e7)")
    p.add_argument("--spmv-reps", type=int, default=50, help="stand-alone SpMV launches timed after the solves")
    return p.parse_args()


def cpu_baseline(args, gpu_nops, gpu_nconv):
    """The oracle (a 'port': Eigen is absent, see oracle/spectra_oracle.hpp) on the host, 1 thread."""
    import numpy as np

    import oracle as O

    t0 = time.time()
    rp, ci, v = O.synth_band_csr(args.n)
    op = O.Op.csr(args.n, args.n, rp, ci, v)
    t_gen = time.time() - t0
    secs, nops = O.time_lanczos_steps(op, args.ncv, args.cpu_steps)
    x = O.simple_random(args.n, 0)
    t_spmv = op.time_op(x, 3)
    per_op = secs / nops
    est_total = per_op * gpu_nops
    return {
        "value": gpu_nconv / est_total,
        "unit": "eigenpairs/s",
        "cores": 1,
        "kind": "port",
        "sample": (f"oracle init() + {args.cpu_steps} Lanczos steps at n={args.n} ({nops} perform_op, {secs:.1f} s, "
                   f"{per_op:.3f} s/op incl. re-orthogonalisation; matrix generation {t_gen:.1f} s not timed), "
                   f"extrapolated to the GPU run's {gpu_nops} operations — early steps orthogonalise against few "
                   f"columns, so this favours the CPU"),
        "seconds_per_op": per_op,
        "spmv_seconds": t_spmv,
        "spmv_gbps": (12.0 * len(v) + 20.0 * args.n + 4) / t_spmv / 1e9,
    }


def pmc_traffic(n, fmt):
    """HBM bytes per launch of the fused SpMV (the instantiation the solve used: 0 int32 indices, 1 offset codes,
    2 diagonal storage) as measured by the committed PMC passes (tools/pmc_summarize.py), or None."""
    import glob

    here = os.path.dirname(os.path.abspath(__file__))
    for path in sorted(glob.glob(os.path.join(here, "profiles", "*pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            if int(d.get("n", -1)) != int(n):
                continue
            for name, rec in d["kernels"].items():
                args_ = name.split("<", 1)[1].rstrip(">").split(",") if "<" in name else []
                is_coded = len(args_) >= 4 and args_[3].strip() == "true"
                if fmt == 2 and name.startswith("k_spmv_dia") and "<true" in name:
                    return float(rec["hbm_bytes"])
                if fmt != 2 and name.startswith("k_spmv_csr_stream<true") and is_coded == (fmt == 1):
                    return float(rec["hbm_bytes"])
        except Exception:  # noqa: BLE001 - a malformed summary just means "no PMC figure"
            continue
    return None


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    import spectra_amd as sa
    from spectra_amd import dist as sdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or args.gpus > 1 or os.environ.get("MISPEC_FORCE_COMM") == "1":
        rank, world = sdist.init_process_group("nccl")
        assert world == args.gpus, f"launched {world} ranks for --gpus {args.gpus}"
        using_dist = True
    else:
        rank, world = 0, 1
        using_dist = False
    local = int(os.environ.get("LOCAL_RANK", "0"))
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

    def solve(profile):
        eigs = sa.SymEigsSolver(op, args.nev, args.ncv)
        if profile:
            eigs.profile(profile)
        eigs.init()
        nconv = eigs.compute(rule, 1000, args.tol)
        ncols = eigs.eigenvectors(to_host=False)  # V * Y formed in HBM (1.6 GB at n = 1e7; not pulled over PCIe)
        return eigs, nconv, ncols

    exchange_note = None
    if world > 1:
        # Self-check of the point-to-point neighbour exchange (outside the timed region): if the solve it drives
        # does not reach the residual bar on every rank, every rank falls back to the plain all-gather.
        chk, nconv_chk, _ = solve(False)
        r = chk.residuals()
        bad = int(nconv_chk < args.nev or not np.all(np.isfinite(r)) or float(r.max()) > 1e-8)
        flag = torch.tensor([bad], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) and chk.exchange_info()[0]:
            os.environ["MISPEC_EXCHANGE"] = "allgather"
            exchange_note = "neighbour exchange failed its residual self-check; all-gather used"
        del chk
    for _ in range(args.warmup):
        solve(False)
    barrier()
    t0 = time.perf_counter()
    solvers = []
    total_pairs = 0
    for _ in range(args.steps):
        # level 2: HIP events bracket only the operator applications (the roofline figure is measured live in the
        # timed region); the other families would cost ~10 more event records per Lanczos step
        eigs, nconv, ncols = solve(0 if args.no_profile else args.profile_level)
        total_pairs += nconv
        solvers.append(eigs)
    barrier()
    elapsed = sdist.max_over_ranks(time.perf_counter() - t0)

    # ---- everything below is outside the timed region -------------------------------------------------
    eigs = solvers[-1]
    prof = {k: 0.0 for k in eigs.get_profile()}
    for s in solvers:
        for k, v in s.get_profile().items():
            prof[k] = prof[k] + v if k != "spmv_bytes" else v
    # per-family kernel split: one more solve with every family instrumented, not part of `value`
    split = None
    if not args.no_profile:
        full, _, _ = solve(1)
        split = full.get_profile()
        del full
    resid = eigs.residuals()
    evals = eigs.eigenvalues()
    spmv_ms = prof["ms_spmv"] / max(prof["n_spmv"], 1)
    spmv_bytes = prof["spmv_bytes"]
    achieved = spmv_bytes / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0

    # The timed region runs whatever storage format the library picked for this matrix (diagonal storage for the
    # benchmark's band matrix).  BASELINE.json words its roofline target for a CSR SpMV, so the two CSR kernels of the same
    # matrix are measured as well — one extra solve each, outside the timed region, same dispatch-bound events.
    csr_kernels = None
    if world == 1 and not args.no_profile and op.spmv_format() != 0:
        csr_kernels = {}
        try:
            for fmt, name in ((1, "csr_offset_codes"), (0, "csr_int32")):
                op.set_spmv_format(fmt)
                if op.spmv_format() != fmt:
                    continue
                alt, alt_nconv, _ = solve(2)
                pa = alt.get_profile()
                ms = pa["ms_spmv"] / max(pa["n_spmv"], 1)
                gbps = spmv_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                csr_kernels[name] = {"ms_per_launch": ms, "achieved": gbps, "frac": gbps / HBM_PEAK_GBPS, "launches": int(pa["n_spmv"]),
                                     "nconv": int(alt_nconv), "num_operations": int(alt.num_operations())}
                del alt
        finally:
            op.set_spmv_format(-1)

    # stand-alone SpMV (same kernel, x resident) as a cross-check of the in-loop number
    x = torch.rand(args.n if world == 1 else int(sa.lib().mispec_shard_block(args.n, world)) * world, dtype=torch.float64,
                   device="cuda") - 0.5
    y = torch.empty(op.local_rows() + 2, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    op.spmv_time(x.data_ptr(), y.data_ptr(), 5)
    alone_ms = op.spmv_time(x.data_ptr(), y.data_ptr(), args.spmv_reps)

    halo, recv_doubles = eigs.exchange_info()
    if world == 1:
        exchange_desc = ""
    elif halo:
        exchange_desc = (f", RCCL point-to-point exchange of the referenced parts of the Krylov vector per SpMV "
                         f"({recv_doubles * 8 / 1e6:.2f} MB received by rank 0; the all-gather would move "
                         f"{(world - 1) * int(sa.lib().mispec_shard_block(args.n, world)) * 8 / 1e6:.1f} MB)")
    else:
        exchange_desc = ", RCCL all-gather of the Krylov vector per SpMV" + (f" ({exchange_note})" if exchange_note else "")
    if rank == 0:
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
            },
            "roofline": {
                "kernel": ("k_spmv_dia_win / k_spmv_dia" if op.spmv_format() == 2 else "k_spmv_csr_stream") + " (SpMV fused with w -= beta*v_prev and the alpha dot)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": pmc_traffic(args.n, op.spmv_format()) if world == 1 else None,
                "traffic_source": "bytes per launch of the in-loop SpMV from the newest profiles/*pmc_traffic.json "
                                  "(rocprofv3 PMC passes need their own profiler run; see profiles/README.md)",
                "bytes_per_launch": spmv_bytes,
                "bytes_note": "algorithmic bytes of a CSR SpMV with int32 indices: 12 nnz + 4 (rows+1) + 8 cols + 8 rows (SURVEY.md 8d)",
                # what this matrix's index format makes the kernel move at least (x once): with offset codes the column
                # index costs 1 byte instead of 4, so `achieved` can exceed what the same time buys in raw HBM bytes
                "index_format": {0: "CSR, int32 column indices",
                                 1: f"CSR, offset codes: 1 byte per entry into {op.offset_codes()} diagonals",
                                 2: f"diagonal storage: {op.offset_codes()} diagonals, no index, no gather"}[op.spmv_format()],
                "stored_bytes_per_launch": op.stored_bytes(),
                "stored_gbps": op.stored_bytes() / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0,
                "ms_per_launch": spmv_ms,
                "launches": int(prof["n_spmv"]),
                # the launch that is timed also does w -= beta*v_prev and the <v, w> partials (Lanczos.h:139,142): two more
                # vector streams (v_prev, v) that the BASELINE.md formula above does not credit
                "fused_epilogue_bytes_per_launch": spmv_bytes + 16.0 * op.local_rows(),
                "fused_epilogue_frac": (spmv_bytes + 16.0 * op.local_rows()) / (spmv_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if spmv_ms > 0 else 0.0,
                "csr_kernels_same_matrix": csr_kernels,
                "csr_kernels_note": "in-loop figures of the CSR SpMV kernels on the same matrix (one extra solve each, outside the timed region); "
                                    "null when the timed region already ran the int32 CSR kernel",
                "standalone_ms_per_launch": alone_ms,
                "standalone_gbps": spmv_bytes / (alone_ms * 1e-3) / 1e9,
            },
            "solve": {
                "nconv": int(total_pairs // args.steps), "num_operations": int(eigs.num_operations()),
                "num_iterations": int(eigs.num_iterations()), "max_residual": float(resid.max()) if len(resid) else None,
                "lambda_max": float(evals.max()) if len(evals) else None, "lambda_min": float(evals.min()) if len(evals) else None,
                "host_syncs_per_solve": prof["n_host_sync"] / args.steps,
            },
            "kernels_ms_per_solve": ({k[3:]: split[k] for k in split if k.startswith("ms_")} if split else None),
            "kernels_ms_note": "from one additional solve with every kernel family bracketed by HIP events, outside the timed region",
            "kernels_launches_per_solve": {k[2:]: prof[k] / args.steps for k in prof if k.startswith("n_")},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, int(eigs.num_operations()), int(total_pairs // args.steps))
        print(json.dumps(out), flush=True)
    if using_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
