"""Row-sharded path exercised on ONE GPU: `world` host threads, each with its own context,
stream and row shard, joined by the library's loopback communicator (all-gather of the Krylov vector before
every SpMV, sum all-reduce of alpha / |f|^2 / V'f).  The SPMD code is the one a multi-process RCCL run
executes; only the transport differs."""
import ctypes as C
import threading

import numpy as np
import pytest

import oracle as O
import spectra_amd as sa
from spectra_amd import _capi

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["onesweep", "onesweep-two-reductions", "reference"], autouse=True)
def orth_env(request, monkeypatch):
    """Every test of this module runs under the defaults of the orthogonalisation scheme (MISPEC_ORTH: the library default
    `onesweep` — one all-reduce per lagged step since round 5 —, the same steps with the separate alpha reduction of round 4,
    MISPEC_ONE_REDUCTION=0, and the reference's two-pass control flow); solvers that set a mode themselves are run once."""
    params = getattr(getattr(request.node, "callspec", None), "params", {})
    if "orth" in params and request.param != "onesweep":
        pytest.skip("this test selects its modes itself")
    monkeypatch.setenv("MISPEC_ORTH", "reference" if request.param == "reference" else "onesweep")
    monkeypatch.setenv("MISPEC_ONE_REDUCTION", "0" if request.param == "onesweep-two-reductions" else "1")
    return request.param


def run_sharded(world, n, offsets, nev, ncv, rule, tol, exchange=None, orth=None, keep_vectors=True, profile=False, make_op=None,
                spmv_format=None):
    import os

    lib = sa.lib()
    old_env = os.environ.pop("MISPEC_EXCHANGE", None)
    if exchange:
        os.environ["MISPEC_EXCHANGE"] = exchange  # read when the factorisation plans its exchange
    grp = C.c_void_p()
    _capi.check(lib.mispec_loopback_create(world, C.byref(grp)))
    results, errors = [None] * world, []

    def worker(rank):
        try:
            ctx = sa.Context(0)
            _capi.check(lib.mispec_loopback_attach(grp, ctx.h, rank))
            ctx.rank, ctx.world = rank, world
            if make_op is not None:  # a matrix from host arrays: every rank is handed the whole matrix and keeps its rows
                op = make_op(ctx)
            else:
                op = sa.SparseSymMatProd.synth_band(n, offsets=offsets, ctx=ctx) if offsets is not None else \
                    sa.SparseSymMatProd.synth_band(n, ctx=ctx)
            chosen = op.spmv_format()
            if spmv_format is not None:
                op.set_spmv_format(spmv_format)
            eigs = sa.SymEigsSolver(op, nev, ncv)
            if orth is not None:  # None: the library default / MISPEC_ORTH
                eigs.set_orth_mode(orth)
            if profile:
                eigs.profile(1)
            eigs.init()
            nconv = eigs.compute(rule, 1000, tol)
            results[rank] = dict(nconv=nconv, info=eigs.info(), evals=eigs.eigenvalues(),
                                 X=eigs.eigenvectors() if keep_vectors else None,
                                 nops=eigs.num_operations(), niter=eigs.num_iterations(), res=eigs.residuals(),
                                 rows=sa.shard_range(n, world, rank), local=op.local_rows(), exchange=eigs.exchange_info(),
                                 overlap=eigs.overlap_info(), orth=eigs.orth_info(), chosen_format=chosen,
                                 format=op.spmv_format(),
                                 profile=eigs.get_profile() if profile else None)
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=1200)
    os.environ.pop("MISPEC_EXCHANGE", None)
    if old_env is not None:
        os.environ["MISPEC_EXCHANGE"] = old_env
    assert not errors, errors
    _capi.check(lib.mispec_loopback_destroy(grp))
    return results


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_equals_unsharded(ctx, world):
    n, offsets, nev, ncv = 50_001, (1, 2, 3, 100, 101, 5000, 5001), 8, 24
    single = sa.SymEigsSolver(sa.SparseSymMatProd.synth_band(n, offsets=offsets, ctx=ctx), nev, ncv)
    single.init()
    assert single.compute(sa.SortRule.LargestAlge, 1000, 1e-11) == nev
    ev1, X1 = single.eigenvalues(), single.eigenvectors()

    res = run_sharded(world, n, offsets, nev, ncv, sa.SortRule.LargestAlge, 1e-11)
    for r in res:
        assert r["nconv"] == nev and r["info"] == sa.CompInfo.Successful
        assert np.array_equal(r["evals"], res[0]["evals"])          # every rank holds the same H
        assert r["nops"] == res[0]["nops"] and r["niter"] == res[0]["niter"]
        assert r["local"] == r["rows"][1] - r["rows"][0] == r["X"].shape[0]
        assert r["res"].max() <= 1e-10
    assert np.abs(res[0]["evals"] - ev1).max() < 1e-10
    X = np.vstack([r["X"] for r in res])                              # row blocks tile [0, n)
    assert X.shape == X1.shape
    assert np.abs(np.abs(np.sum(X * X1, axis=0)) - 1.0).max() < 1e-8  # same vectors up to sign
    rp, ci, v = O.synth_band_csr(n, offsets=offsets)
    import scipy.sparse as sp
    A = sp.csr_matrix((v, ci, rp), shape=(n, n))
    assert (np.linalg.norm(A @ X - X * res[0]["evals"], axis=0) / np.linalg.norm(X, axis=0)).max() <= 1e-10


@pytest.mark.parametrize("world", [2, 5])
def test_neighbour_exchange_equals_allgather(world):
    # A banded matrix references only a halo of the neighbouring shards: the plan must pick the point-to-point
    # exchange, move far fewer doubles than the all-gather, and change nothing in the results (same x values reach
    # the same SpMV) — bit-identical eigenvalues and counters.
    n, offsets, nev, ncv = 40_003, (1, 2, 3, 50, 51, 1500, 1501), 6, 20
    halo = run_sharded(world, n, offsets, nev, ncv, sa.SortRule.LargestMagn, 1e-11)
    full = run_sharded(world, n, offsets, nev, ncv, sa.SortRule.LargestMagn, 1e-11, exchange="allgather")
    block = sa.shard_block(n, world)
    for rank, (h, f) in enumerate(zip(halo, full)):
        assert h["exchange"][0] and not f["exchange"][0]
        neighbours = (rank > 0) + (rank < world - 1)
        assert h["exchange"][1] == 1501 * neighbours
        assert f["exchange"][1] == block * (world - 1)
        assert h["nconv"] == f["nconv"] == nev
        assert np.array_equal(h["evals"], f["evals"]) and np.array_equal(h["X"], f["X"])
        assert (h["nops"], h["niter"]) == (f["nops"], f["niter"])


def test_wide_pattern_keeps_the_allgather():
    # couplings across half the matrix: every rank reads most of every other slice -> one all-gather is kept
    n, world = 30_000, 3
    res = run_sharded(world, n, (1, 7000, 14000), 4, 16, sa.SortRule.LargestAlge, 1e-10)
    forced = run_sharded(world, n, (1, 7000, 14000), 4, 16, sa.SortRule.LargestAlge, 1e-10, exchange="halo")
    for r, f in zip(res, forced):
        assert not r["exchange"][0] and f["exchange"][0]
        assert np.array_equal(r["evals"], f["evals"]) and r["nops"] == f["nops"]
        assert r["res"].max() <= 1e-9


def test_sharded_wide_basis(ctx):
    # ncv > 64 on two shards: panelled orthogonalisation + one all-reduce of the whole record per reduction
    n, offsets, nev, ncv = 30_001, (1, 2, 3, 100, 101, 2000, 2001), 34, 80
    single = sa.SymEigsSolver(sa.SparseSymMatProd.synth_band(n, offsets=offsets, ctx=ctx), nev, ncv)
    single.init()
    assert single.compute(sa.SortRule.LargestMagn, 1000, 1e-11) == nev
    res = run_sharded(2, n, offsets, nev, ncv, sa.SortRule.LargestMagn, 1e-11)
    for r in res:
        assert r["nconv"] == nev and np.array_equal(r["evals"], res[0]["evals"])
        assert r["res"].max() <= 1e-10
    assert np.abs(res[0]["evals"] - single.eigenvalues()).max() < 1e-10


@pytest.mark.parametrize("world,exchange", [(2, None), (3, None), (2, "allgather")])
def test_exchange_overlapped_with_the_local_rows_changes_nothing(world, exchange):
    # The row-blocks that read only the rank's own slice are multiplied while the exchange runs on a second stream
    # (SURVEY.md 8e); same kernels on the same blocks => bit-identical to the run without overlap (MISPEC_OVERLAP=0).
    import os

    n, offsets, nev, ncv = 120_001, (1, 2, 3, 50, 51, 1500, 1501), 6, 20
    with_overlap = run_sharded(world, n, offsets, nev, ncv, sa.SortRule.LargestMagn, 1e-11, exchange=exchange)
    os.environ["MISPEC_OVERLAP"] = "0"
    try:
        without = run_sharded(world, n, offsets, nev, ncv, sa.SortRule.LargestMagn, 1e-11, exchange=exchange)
    finally:
        del os.environ["MISPEC_OVERLAP"]
    for a, b in zip(with_overlap, without):
        first, count, total = a["overlap"]
        assert count >= total - 2 * 7 and count < total       # all but the blocks within 1501 rows of a shard boundary
        assert b["overlap"][1] == 0
        assert a["nconv"] == b["nconv"] == nev
        assert np.array_equal(a["evals"], b["evals"]) and np.array_equal(a["X"], b["X"])
        assert (a["nops"], a["niter"]) == (b["nops"], b["niter"])
        assert a["res"].max() <= 1e-10


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("orth", ["onesweep", "onesweep-eager", "onesweep-recorrect", "reference"])
def test_sharded_device_run_that_stops_mid_sweep(ctx, world, orth):
    # ncv = 60 on a small band: Ritz pairs converge inside a sweep, the corrections grow beyond what a lagged step may carry
    # and the device-driven run STOPS in mid-sweep (state_stops); the host continues from the record of the stopping pass while
    # the launches still in the queue — their reductions and all-reduces included — must leave that record alone.  Sharded,
    # only ONE correction is enqueued speculatively, so a record overwritten by a later no-op step would not be repaired by
    # chance.  Every rank identical, equal to the unsharded solve, residuals and orthogonality at rounding level.
    n, offsets, nev, ncv = 20_001, (1, 2, 3, 50, 51, 1500, 1501), 10, 60
    single = sa.SymEigsSolver(sa.SparseSymMatProd.synth_band(n, offsets=offsets, ctx=ctx), nev, ncv)
    single.set_orth_mode(orth)
    single.init()
    assert single.compute(sa.SortRule.LargestAlge, 1000, 1e-11) == nev
    res = run_sharded(world, n, offsets, nev, ncv, sa.SortRule.LargestAlge, 1e-11, orth=orth)
    X = np.vstack([r["X"] for r in res])
    for r in res:
        assert r["nconv"] == nev and r["info"] == sa.CompInfo.Successful
        assert np.array_equal(r["evals"], res[0]["evals"]) and (r["nops"], r["niter"]) == (res[0]["nops"], res[0]["niter"])
        assert r["res"].max() <= 1e-10
        if orth != "reference":
            assert r["orth"]["mode"] == "onesweep" and r["orth"]["state_stops"] > 0
            assert (r["orth"]["fused_restarts"] > 0) == (orth != "onesweep-eager")
            assert (r["orth"]["fused_recorrected"] > 0) == (orth == "onesweep-recorrect")
    assert np.abs(res[0]["evals"] - single.eigenvalues()).max() < 1e-10
    assert abs(res[0]["nops"] - single.num_operations()) <= ncv - nev
    assert np.abs(X.T @ X - np.eye(nev)).max() <= 1e-10


@pytest.mark.parametrize("world,which", [(2, "staged"), (3, "staged"), (2, "tiles")])
def test_scattered_pattern_sharded_uses_the_staged_format(ctx, world, which, monkeypatch):
    # a pattern with uniformly scattered columns, row-sharded: every shard builds the staged (or tile) image of ITS rows — all
    # columns, x = the gathered vector — and the run equals the one from plain CSR on the same shards bit for bit, and the
    # unsharded solve to rounding
    import scipy.sparse as sp

    monkeypatch.setenv("MISPEC_SPMV_STAGED", "auto" if which == "staged" else "0")
    n, nev, ncv = 600_000, 6, 20
    rng = np.random.default_rng(11)
    rows = np.repeat(np.arange(n), 3)
    cols = rng.integers(0, n, rows.size)
    T = sp.coo_matrix((rng.uniform(-1.0, 1.0, rows.size), (rows, cols)), shape=(n, n)).tocsr()
    d = rng.uniform(1.0, 2.0, n)
    d[rng.choice(n, 8, replace=False)] = 50.0 + 5.0 * np.arange(8)  # separated outliers: a short solve
    A = (T + T.T + sp.diags(d)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    L = sp.tril(A).tocsc()
    make = lambda c: sa.SparseSymMatProd(L, ctx=c)
    fmt = 4 if which == "staged" else 3
    res = run_sharded(world, n, None, nev, ncv, sa.SortRule.LargestAlge, 1e-10, make_op=make)
    csr = run_sharded(world, n, None, nev, ncv, sa.SortRule.LargestAlge, 1e-10, make_op=make, spmv_format=0)
    for r, c in zip(res, csr):
        assert r["chosen_format"] == r["format"] == fmt and c["format"] == 0
        assert r["overlap"][1] == 0  # these formats take no row-block sub-ranges: nothing is multiplied ahead of the exchange
        assert r["nconv"] == nev and np.array_equal(r["evals"], c["evals"]) and np.array_equal(r["X"], c["X"])
        assert (r["nops"], r["niter"]) == (c["nops"], c["niter"])
    single = sa.SymEigsSolver(sa.SparseSymMatProd(L, ctx=ctx), nev, ncv)
    single.init()
    assert single.compute(sa.SortRule.LargestAlge, 1000, 1e-10) == nev
    assert np.abs(single.eigenvalues() - res[0]["evals"]).max() < 1e-9
    X = np.vstack([r["X"] for r in res])
    assert (np.linalg.norm(A @ X - X * res[0]["evals"], axis=0) / np.linalg.norm(X, axis=0)).max() <= 1e-8
