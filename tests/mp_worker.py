"""One rank of a multi-PROCESS row-sharded solve (started by tests/test_gpu_multiproc.py or tools/ through
torch.distributed.run).  All ranks may share device 0 (MP_DEVICE=shared) — then the transport is "gloo-staged" (collectives
staged through host memory, spectra_amd.dist.HostStagedComm) — or own one GPU each with the RCCL transport.

    MP_OUT=dir MP_N=40003 MP_OFFSETS=1,2,3,50,51,1500,1501 MP_NEV=6 MP_NCV=20 MP_RULE=LargestMagn MP_TRANSPORT=gloo-staged|rccl|torch
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    import torch
    import torch.distributed as dist

    import spectra_amd as sa
    from spectra_amd import dist as sdist

    transport = os.environ.get("MP_TRANSPORT", "gloo-staged")
    shared = os.environ.get("MP_DEVICE", "shared") == "shared"
    rank, world = sdist.init_process_group("gloo" if transport == "gloo-staged" else "nccl")
    device = 0 if shared else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(device)
    ctx = sdist.make_context(device, transport=transport)
    n = int(os.environ.get("MP_N", "40003"))
    offsets = tuple(int(x) for x in os.environ.get("MP_OFFSETS", "1,2,3,50,51,1500,1501").split(","))
    nev, ncv = int(os.environ.get("MP_NEV", "6")), int(os.environ.get("MP_NCV", "20"))
    rule = sa.SortRule[os.environ.get("MP_RULE", "LargestMagn")]
    op = sa.SparseSymMatProd.synth_band(n, offsets=offsets, ctx=ctx)
    eigs = sa.SymEigsSolver(op, nev, ncv)
    if os.environ.get("MP_ORTH"):  # unset: the library default / MISPEC_ORTH
        eigs.set_orth_mode(os.environ["MP_ORTH"])
    eigs.init()
    nconv = eigs.compute(rule, 1000, 1e-11)
    # MP_SAVE_X=0 (full-size runs): keep the eigenvectors out of the result files, hand back per-column checksums instead
    save_x = os.environ.get("MP_SAVE_X", "1") != "0"
    X = eigs.eigenvectors()
    xsum = np.array([float(np.sum(X[:, j] * X[:, j])) for j in range(X.shape[1])])
    if not save_x:
        X = np.zeros((0, X.shape[1]))
    out = os.environ["MP_OUT"]
    np.savez(os.path.join(out, f"rank{rank}.npz"), nconv=nconv, info=int(eigs.info()), evals=eigs.eigenvalues(), X=X,
             nops=eigs.num_operations(), niter=eigs.num_iterations(), res=eigs.residuals(), rows=np.array(sa.shard_range(n, world, rank)),
             xsum=xsum,
             exchange=np.array(eigs.exchange_info(), dtype=np.int64))
    with open(os.path.join(out, f"rank{rank}.json"), "w") as f:
        json.dump({"rank": rank, "world": world, "transport": transport, "device": device, "pid": os.getpid(), "nconv": int(nconv),
                   "nops": int(eigs.num_operations()), "halo": bool(eigs.exchange_info()[0])}, f)
    del eigs, op
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
