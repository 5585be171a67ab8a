"""N > 1 plumbing on CPU: two `gloo` ranks (127.0.0.1) exercise the product's distributed helpers
(spectra_amd.dist: process-group setup, TorchComm's two collectives, the equal-block row partition, result
assembly, max-over-ranks timing) and replay the sharded Lanczos PROTOCOL — what is all-gathered and what is
all-reduced each step, with the padded block layout the C++ path uses — in numpy, checking it against the
unsharded CPU oracle.  No GPU kernels run here (there is no CPU fallback to run); the same SPMD code with
real kernels is covered on the GPU by tests/test_gpu_sharded.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, m, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import scipy.sparse as sp
    import torch

    import oracle as O
    import spectra_amd as sa
    from spectra_amd import dist as sdist

    r, w = sdist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    comm = sdist.TorchComm()
    blk = sa.lib().mispec_shard_block(n, world)
    b, e = sa.shard_range(n, world, rank)
    nloc = e - b

    # --- collectives of the adapter ----------------------------------------------------------------
    send = torch.full((blk,), float(rank + 1), dtype=torch.float64)
    recv = torch.zeros(blk * world, dtype=torch.float64)
    comm.allgather(send, recv)
    assert all(torch.all(recv[q * blk:(q + 1) * blk] == q + 1) for q in range(world))
    red = torch.tensor([1.0, float(rank)], dtype=torch.float64)
    comm.allreduce_sum(red)
    assert red.tolist() == [float(world), float(sum(range(world)))]
    assert sdist.max_over_ranks(0.5 + rank) == 0.5 + (world - 1)

    # --- the sharded Lanczos protocol on this rank's rows -------------------------------------------
    offsets = (1, 2, 3, 50, 51)
    rp, ci, v = O.synth_band_csr(n, offsets=offsets)
    A = sp.csr_matrix((v, ci, rp), shape=(n, n))
    Aloc = A[b:e]                      # global column indices, like the device shard
    v0 = O.simple_random(n, 0)
    assert np.array_equal(sdist.local_rows_of(v0), v0[b:e])

    def gather(x_loc):                 # all-gather of equal padded blocks == position of global row i is i
        s = torch.zeros(blk, dtype=torch.float64)
        s[:nloc] = torch.from_numpy(np.ascontiguousarray(x_loc))
        full = torch.zeros(blk * world, dtype=torch.float64)
        comm.allgather(s, full)
        return full.numpy()[:n]

    def allsum(vals):
        t = torch.tensor(np.atleast_1d(np.asarray(vals, dtype=np.float64)))
        comm.allreduce_sum(t)
        return t.numpy()

    V = np.zeros((nloc, m))
    H = np.zeros((m, m))
    x = Aloc @ gather(v0[b:e])                                   # Arnoldi.h:153
    V[:, 0] = x / np.sqrt(allsum(x @ x)[0])
    wv = Aloc @ gather(V[:, 0])
    H[0, 0] = allsum(V[:, 0] @ wv)[0]
    f = wv - V[:, 0] * H[0, 0]
    beta = np.sqrt(allsum(f @ f)[0])
    eps = np.finfo(float).eps
    for i in range(1, m):                                        # Lanczos.h:88-183, no restarts expected here
        V[:, i] = f / beta
        H[i, i - 1] = H[i - 1, i] = beta
        wv = Aloc @ gather(V[:, i]) - beta * V[:, i - 1]
        H[i, i] = allsum(V[:, i] @ wv)[0]
        f = wv - H[i, i] * V[:, i]
        red = allsum(np.concatenate([V[:, :i + 1].T @ f, [f @ f]]))     # one fused reduction: V'f and |f|^2
        Vf, beta = red[:-1], np.sqrt(red[-1])
        count = 0
        while count < 5 and np.abs(Vf).max() > eps * beta:
            f = f - V[:, :i + 1] @ Vf
            H[i - 1, i] += Vf[i - 1]
            H[i, i - 1] = H[i - 1, i]
            H[i, i] += Vf[i]
            red = allsum(np.concatenate([V[:, :i + 1].T @ f, [f @ f]]))
            Vf, beta = red[:-1], np.sqrt(red[-1])
            count += 1

    Vfull = sdist.gather_rows(V, n)

    # --- the same factorisation by the library's DEFAULT steps: one sweep of V and ONE all-reduce per step ------------------------
    # (device: fac.hip lanczos_step_lagged, krylov.hip k_orth_lagged<…, ONERED> / start_next_onered; CPU
    # restatement: oracle/onesweep_variant.hpp flavour one-reduction).  Per step: all-gather of the UN-normalised residual, product,
    # ONE message = the record of the previous pass ([V, v_{i-1}]'f~, V'v_{i-1} when a correction was pending, |f~|^2) with this
    # rank's <f~, A f~> behind it; then the pass: v_i = (f~ - V c)/beta, w = u/beta - beta v_{i-1}, f~' = w - alpha~ v_i.
    calls = {"allsum": 0}

    def allsum1(vals):
        calls["allsum"] += 1
        return allsum(vals)

    V1 = np.zeros((nloc, m))
    H1 = np.zeros((m, m))
    V1[:, 0] = V[:, 0]
    H1[0, 0] = H[0, 0]
    f1 = Aloc @ gather(V1[:, 0]) - V1[:, 0] * H1[0, 0]
    st = {"beta": np.sqrt(allsum(f1 @ f1)[0]), "pending": False, "c": None, "prev_last": 0.0, "alpha": 0.0}
    rec = None  # the unreduced record of the previous pass: (local sums, i, had_chk)

    def finish(i, red, had_chk):  # the scalar tail of step i (krylov.hip finish_lagged)
        beta_used, c = st["beta"], st["c"]
        if had_chk:
            # Lanczos.h:156 after the lagged correction, in units of beta: beyond eps the device leaves the lagged path for the
            # reference's loop (kStepLagCheck); the replay has no such fall-back and accepts rounding-level excess
            assert np.abs(red[i + 1:2 * i + 1]).max() <= 8 * eps, np.abs(red[i + 1:2 * i + 1]).max() / eps
        H1[i, i] = st["alpha"] - (c[i - 1] if had_chk else 0.0)
        sub = beta_used
        if had_chk:
            sub -= ((H1[i - 1, i - 2] * c[i - 2] if i >= 2 else 0.0) + H1[i - 1, i - 1] * c[i - 1]) / beta_used
        H1[i, i - 1] = H1[i - 1, i] = sub
        Vf, gamma2 = red[:i + 1], red[-1]
        gamma = np.sqrt(gamma2)
        st.update(beta=gamma, pending=False, c=None, prev_last=Vf[i])
        if np.abs(Vf).max() > eps * gamma:  # a correction is needed: it rides on the next pass
            c2 = float(Vf @ Vf)
            assert c2 <= 1e-6 * gamma2
            H1[i - 1, i] += Vf[i - 1]
            H1[i, i - 1] = H1[i - 1, i]
            H1[i, i] += Vf[i]
            st.update(beta=np.sqrt(gamma2 - c2), pending=True, c=Vf.copy())

    for i in range(1, m):
        if rec is None:  # first step: the two-reduction form
            vcol = f1 / st["beta"]
            wv = Aloc @ gather(vcol) - st["beta"] * V1[:, i - 1]
            st["alpha"] = allsum1(vcol @ wv)[0]
            beta_pass, had_chk, c = st["beta"], False, None
        else:
            u = Aloc @ gather(f1)  # the product does not wait for beta
            loc, ip, chk_prev = rec
            msg = allsum1(np.concatenate([loc, [f1 @ u]]))  # ONE all-reduce: the previous pass's record + <f~, A f~>
            finish(ip, msg[:-1], chk_prev)
            beta_pass, had_chk, c = st["beta"], st["pending"], st["c"]
            st["alpha"] = msg[-1] / beta_pass ** 2 - st["prev_last"]
            wv = u / beta_pass - beta_pass * V1[:, i - 1]
        vi = f1 - V1[:, :i] @ c if had_chk else f1
        V1[:, i] = vi / beta_pass
        chk = V1[:, :i].T @ V1[:, i] if had_chk else np.zeros(0)
        f1 = wv - st["alpha"] * V1[:, i]
        rec = (np.concatenate([V1[:, :i + 1].T @ f1, chk, [f1 @ f1]]), i, had_chk)
    loc, ip, chk_prev = rec
    finish(ip, allsum1(loc), chk_prev)  # the last record of the sweep is reduced at once
    if st["pending"]:  # ... and its correction applied (on the device it rides on the restart's V*Q pass)
        f1 = f1 - V1 @ st["c"]
    assert calls["allsum"] == m  # one per step (m - 1 steps) + the sweep's last record
    V1full = sdist.gather_rows(V1, n)
    if rank == 0:
        fac = O.Factorization(O.Op.csr(n, n, rp, ci, v), m, True)
        fac.init(v0)
        fac.factorize_from(1, m)
        V0, H0, f0 = fac.matrices()
        np.savez(os.path.join(out_dir, "result.npz"), dH=np.abs(H - H0).max(), dV=np.abs(Vfull - V0).max(),
                 orth=np.abs(Vfull.T @ Vfull - np.eye(m)).max(),
                 dH1=np.abs(H1 - H0).max(), dV1=np.abs(V1full - V0).max(), orth1=np.abs(V1full.T @ V1full - np.eye(m)).max())
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1001, 4000])
def test_two_rank_gloo(tmp_path, n):
    import torch.multiprocessing as mp

    world, m = 2, 10
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, m, str(tmp_path)), nprocs=world, join=True)
    res = np.load(tmp_path / "result.npz")
    assert res["dH"] < 1e-10 and res["dV"] < 1e-9 and res["orth"] < 1e-12
    # the one-sweep / one-reduction protocol: the same factorisation to rounding, from one all-reduce per step
    assert res["dH1"] < 1e-10 and res["dV1"] < 1e-9 and res["orth1"] < 1e-12
