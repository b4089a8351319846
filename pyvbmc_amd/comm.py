"""Multi-GPU plumbing: one process per GPU, one RCCL communicator, ONE collective
per ELBO evaluation.

The reference has no communication layer.  Here the Monte-Carlo sample rows of
every mixture component are split evenly over the ranks (``shard_rows``), each
rank's kernels produce the raw entropy accumulator of its rows, and a single
``ncclAllReduce(sum, float64)`` of that 1+D*K+2K+D vector over xGMI (inside
libvbmc_hip.so, on the ctx stream) gives every rank the full-job value.  G and the
lower-bound entropy are tiny and replicated.

Rendezvous: the 128-byte RCCL unique id travels from rank 0 to the others through
a file (single node, as launched by ``python -m torch.distributed.run``); no
PyTorch is imported.  ``GlooGroup`` is a host-side stand-in with the same
``allreduce_sum`` used by the CPU tests (torch.distributed / gloo) to check the
sharding arithmetic without GPUs.
"""
import os
import time
from pathlib import Path

import numpy as np


def shard_rows(n_half, rank, world):
    """Rows [begin, end) of every component's antithetic half owned by ``rank``.
    Keeps each +/- pair on one rank and the per-rank work equal to within one row."""
    begin = n_half * rank // world
    end = n_half * (rank + 1) // world
    return begin, end


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def _rendezvous_path():
    key = "_".join(
        [
            os.environ.get("MASTER_ADDR", "127.0.0.1").replace(":", "-"),
            os.environ.get("MASTER_PORT", "0"),
            os.environ.get("TORCHELASTIC_RUN_ID", "none"),
            str(os.getppid()),
        ]
    )
    return Path(os.environ.get("VBMC_RDZV_DIR", "/tmp")) / f"vbmc_rdzv_{key}.uid"


def exchange_unique_id(rank, world, make_id, timeout=300.0):
    """Rank 0 creates the id and publishes it atomically; the others poll for it."""
    path = _rendezvous_path()
    t0 = time.time()
    if rank == 0:
        uid = make_id()
        tmp = path.with_suffix(".tmp%d" % os.getpid())
        tmp.write_bytes(uid)
        os.replace(tmp, path)
        return uid
    while True:
        try:
            st = path.stat()
            # ignore leftovers of an earlier launch that reused the same key
            if st.st_size == 128 and st.st_mtime >= t0 - 120.0:
                return path.read_bytes()
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError(f"rank {rank}: no RCCL unique id at {path} after {timeout}s")
        time.sleep(0.02)


def init_from_env(ctx):
    """Join the RCCL communicator described by RANK / WORLD_SIZE (no-op for world 1)."""
    from . import _lib

    rank, world, _ = env_rank_world()
    if world <= 1:
        return rank, world
    uid = exchange_unique_id(rank, world, _lib.comm_unique_id)
    ctx.comm_init(uid, rank, world)
    ctx.comm_barrier()
    if rank == 0:
        try:
            _rendezvous_path().unlink()
        except OSError:
            pass
    return rank, world


class GlooGroup:
    """Host float64 all-reduce over torch.distributed (gloo) -- CPU tests only."""

    def __init__(self):
        import torch.distributed as dist

        self._dist = dist
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()

    def allreduce_sum(self, vec):
        import torch

        t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64).copy())
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return t.numpy()
