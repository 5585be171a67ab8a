"""Row-sharded runs: one process per GPU, `torch.distributed` for rendezvous, RCCL over xGMI for the data path.

The eigensolver needs exactly two collectives: an all-gather of the current Krylov vector
before every SpMV and a sum all-reduce of a few doubles (alpha, |f|^2, V'f) after the reductions.  Two
transports are offered, both ending in RCCL:

  * "rccl"  (default) — the library's own RCCL communicator (`mispec_ctx_set_comm_rccl`): rank 0 creates an
    ncclUniqueId, `torch.distributed` broadcasts it, every rank calls ncclCommInitRank.  Collectives are
    enqueued by C++ on the solver's stream, no Python in the loop.
  * "gloo-staged" — for tests: N processes that may SHARE a device, collectives staged through host memory over gloo
    (HostStagedComm); exercises the real multi-process SPMD path where RCCL cannot build a communicator.
  * "torch" — Python callbacks that run `dist.all_gather_into_tensor` / `dist.all_reduce` (backend "nccl" is
    RCCL on ROCm; "gloo" works for CPU tensors in the tests) on tensors aliasing the library's device buffers.
    The context must then run on torch's current stream so that torch orders the collectives with the kernels.

`torch` is plumbing here (process group, streams); all arithmetic stays in libmispec.so.
"""
import os

import numpy as np

from . import Context, MispecError, shard_range


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / MASTER_*)."""
    import torch
    import torch.distributed as dist

    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1 and "MASTER_ADDR" not in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


class _DeviceArray:
    """Zero-copy view of `count` doubles at a raw device pointer, consumable by torch.as_tensor."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class TorchComm:
    """The two collectives on torch tensors (any backend)."""

    def __init__(self, group=None, stream=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.stream = stream  # torch.cuda.Stream the solver runs on (device collectives are ordered against it)
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def allgather(self, send, recv):
        """recv[r*len(send):(r+1)*len(send)] = send of rank r."""
        self.dist.all_gather_into_tensor(recv, send, group=self.group)

    def allreduce_sum(self, buf):
        self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, group=self.group)

    def _on(self, stream):
        """The stream the library passed to the callback (the solver's stream, or its communication stream while the exchange
        overlaps the local rows): the collective must be ordered on THAT stream — mispec_comm's contract, include/mispec.h."""
        import torch

        if stream:
            return torch.cuda.stream(torch.cuda.ExternalStream(int(stream)))
        return torch.cuda.stream(self.stream)

    # raw-pointer forms used as C callbacks (device memory)
    def _allgather_ptr(self, send_ptr, recv_ptr, count, stream):
        import torch

        with self._on(stream):
            send = torch.as_tensor(_DeviceArray(send_ptr, count), device="cuda")
            recv = torch.as_tensor(_DeviceArray(recv_ptr, count * self.world), device="cuda")
            self.allgather(send, recv)
        return 0

    def _allreduce_ptr(self, buf_ptr, count, stream):
        import torch

        with self._on(stream):
            self.allreduce_sum(torch.as_tensor(_DeviceArray(buf_ptr, count), device="cuda"))
        return 0


class HostStagedComm:
    """The collectives for DEVICE buffers over a CPU process group (gloo): every call drains the solver's stream, stages the
    operands through host memory, runs the gloo collective and copies the result back.  Slow by construction — it exists so
    that the product's SPMD code (row shards, exchange plan, batched all-reduces, identical H on every rank) can run as N
    real processes on a box with fewer GPUs than ranks (the ranks then share a device), where RCCL refuses to build a
    communicator.  Includes the personalised exchange, so the neighbour-exchange plan is exercised as well."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    @staticmethod
    def _view(ptr, count):
        import torch

        return torch.as_tensor(_DeviceArray(ptr, count), device="cuda")

    def _sync(self, stream):
        import torch

        torch.cuda.ExternalStream(int(stream)).synchronize() if stream else torch.cuda.synchronize()

    def allgather_ptr(self, send_ptr, recv_ptr, count, stream):
        import torch

        self._sync(stream)
        send = self._view(send_ptr, count).cpu()
        recv = torch.empty(count * self.world, dtype=torch.float64)
        self.dist.all_gather_into_tensor(recv, send, group=self.group)
        self._view(recv_ptr, count * self.world).copy_(recv)
        torch.cuda.synchronize()
        return 0

    def allreduce_ptr(self, buf_ptr, count, stream):
        import torch

        self._sync(stream)
        dev = self._view(buf_ptr, count)
        # fixed rank order => the same bits on every rank (gloo's ring order depends on the rank)
        parts = torch.empty(count * self.world, dtype=torch.float64)
        self.dist.all_gather_into_tensor(parts, dev.cpu(), group=self.group)
        acc = torch.zeros(count, dtype=torch.float64)
        for r in range(self.world):
            acc += parts[r * count:(r + 1) * count]
        dev.copy_(acc)
        torch.cuda.synchronize()
        return 0

    def exchange_ptr(self, send_ptr, send_off, send_count, recv_ptr, recv_off, recv_count, stream):
        import torch

        self._sync(stream)
        ops, landing = [], []
        for p in range(self.world):
            if p == self.rank:
                continue
            if send_count[p] > 0:
                buf = self._view(send_ptr + 8 * send_off[p], send_count[p]).cpu()
                ops.append(self.dist.P2POp(self.dist.isend, buf, p, group=self.group))
            if recv_count[p] > 0:
                buf = torch.empty(recv_count[p], dtype=torch.float64)
                landing.append((recv_off[p], recv_count[p], buf))
                ops.append(self.dist.P2POp(self.dist.irecv, buf, p, group=self.group))
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        for off, cnt, buf in landing:
            self._view(recv_ptr + 8 * off, cnt).copy_(buf)
        torch.cuda.synchronize()
        return 0


def make_context(device=None, transport=None):
    """Context for this rank's GPU with the communicator attached (no-op communicator when world == 1).

    Requires torch.distributed to be initialised when WORLD_SIZE > 1."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    transport = transport or os.environ.get("MISPEC_COMM", "rccl")
    # MISPEC_FORCE_COMM=1 attaches the communicator even for a single rank, so that the RCCL / torch
    # transports can be smoke-tested on a 1-GPU box (collectives over one rank are copies / no-ops).
    force = os.environ.get("MISPEC_FORCE_COMM", "0") == "1" and dist.is_initialized()
    torch.cuda.set_device(device)
    if (world > 1 or force) and transport == "gloo-staged":
        ctx = Context(device)
        comm = HostStagedComm()
        ctx.set_comm_callbacks(rank, world, comm.allgather_ptr, comm.allreduce_ptr, comm.exchange_ptr)
        ctx._comm = comm
    elif (world > 1 or force) and transport == "torch":
        # a dedicated (non-default) torch stream: the solver's kernels and torch's collectives are both ordered
        # against it (the default stream's handle is 0, which the C ABI reads as "create your own stream")
        stream = torch.cuda.Stream(device=device)
        ctx = Context(device, stream=stream.cuda_stream)
        comm = TorchComm(stream=stream)
        ctx.set_comm_callbacks(rank, world, comm._allgather_ptr, comm._allreduce_ptr)
        ctx._comm = comm
    else:
        ctx = Context(device)
        if world > 1 or force:
            # The library's own RCCL communicator (dlopen'd librccl, unique id from rank 0).  It has never met more than one rank
            # inside a build round (1-GPU leases): should its creation fail on ANY rank — every rank learns it through a MIN
            # all-reduce on the process group — all ranks fall back to torch.distributed's collectives (TorchComm) instead of
            # aborting the run.  MISPEC_COMM=rccl-strict keeps the failure fatal.
            ok = 1
            err = None
            payload = [None]
            if rank == 0:
                try:
                    payload = [Context.rccl_unique_id()]
                except Exception as e:  # noqa: BLE001 - every rank still takes part in the broadcast below (no rank may wait alone)
                    err = e
            dist.broadcast_object_list(payload, src=0)
            if payload[0] is None:
                ok = 0
            else:
                try:
                    ctx.set_comm_rccl(rank, world, payload[0])
                except Exception as e:  # noqa: BLE001
                    ok, err = 0, e
            # every rank learns whether ALL ranks have a communicator before anything is decided (in strict mode a rank
            # without one must not carry on as a silent single-rank context).  What this cannot recover from: ncclCommInitRank is
            # itself collective — a rank that fails BEFORE entering it leaves its peers blocked inside it, and the all-reduce below
            # is never reached; the launcher's timeout ends such a run.
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if transport == "rccl-strict":
                if err is not None:
                    raise err
                if int(flag.item()) == 0:
                    raise MispecError("MISPEC_COMM=rccl-strict: the RCCL communicator could not be created on another rank"
                                      + (" (rank 0 has no unique id)" if payload[0] is None else ""))
                return ctx
            if int(flag.item()) == 0:
                import sys

                if rank == 0:
                    print("spectra_amd.dist: the library's RCCL communicator could not be created (%r on this rank); "
                          "falling back to torch.distributed collectives" % (err,), file=sys.stderr)
                del ctx
                stream = torch.cuda.Stream(device=device)
                ctx = Context(device, stream=stream.cuda_stream)
                comm = TorchComm(stream=stream)
                ctx.set_comm_callbacks(rank, world, comm._allgather_ptr, comm._allreduce_ptr)
                ctx._comm = comm
    return ctx


def max_over_ranks(seconds):
    """Wall time of the slowest rank (the bench contract's max-over-ranks)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(seconds)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(local_block, n, group=None):
    """Assemble the row blocks of every rank (each `local_rows x k`, numpy) into the full n x k array on all ranks."""
    import torch.distributed as dist

    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return np.asarray(local_block)
    parts = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, np.asarray(local_block), group=group)
    full = np.vstack(parts)
    assert full.shape[0] == n, (full.shape, n)
    return full


def local_rows_of(global_vector, n=None):
    """This rank's slice of a replicated length-n array (mispec_shard_range)."""
    import torch.distributed as dist

    if not dist.is_initialized():
        return global_vector
    n = len(global_vector) if n is None else n
    b, e = shard_range(n, dist.get_world_size(), dist.get_rank())
    return global_vector[b:e]
