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
    if rank == 0:
        fac = O.Factorization(O.Op.csr(n, n, rp, ci, v), m, True)
        fac.init(v0)
        fac.factorize_from(1, m)
        V0, H0, f0 = fac.matrices()
        np.savez(os.path.join(out_dir, "result.npz"), dH=np.abs(H - H0).max(), dV=np.abs(Vfull - V0).max(),
                 orth=np.abs(Vfull.T @ Vfull - np.eye(m)).max())
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
