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


def _rendezvous_dir():
    """A directory only this user can enter (mode 0700, owned by us, not a symlink): nobody
    else can pre-create or swap the rendezvous file."""
    import stat
    import tempfile

    base = Path(os.environ.get("VBMC_RDZV_DIR", tempfile.gettempdir()))
    d = base / f"vbmc_rdzv_{os.getuid()}"
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError(f"{d} must be a directory owned by uid {os.getuid()} with mode 0700")
    return d


def _rendezvous_path():
    key = "_".join(
        [
            os.environ.get("MASTER_ADDR", "127.0.0.1").replace(":", "-"),
            os.environ.get("MASTER_PORT", "0"),
            os.environ.get("TORCHELASTIC_RUN_ID", "none"),
            str(os.getppid()),
        ]
    )
    return _rendezvous_dir() / f"{key}.uid"


def _launcher_start_time():
    """Start time (epoch seconds) of the parent process -- the launcher all local ranks share.
    A rendezvous file older than the launcher is a leftover of an earlier launch that happened
    to reuse the same key (port, run id, recycled pid) and must not be joined."""
    try:
        fields = Path(f"/proc/{os.getppid()}/stat").read_text().rsplit(")", 1)[1].split()
        ticks = int(fields[19])  # starttime, field 22 of /proc/<pid>/stat
        btime = next(int(l.split()[1]) for l in Path("/proc/stat").read_text().splitlines() if l.startswith("btime"))
        return btime + ticks / os.sysconf("SC_CLK_TCK")
    except (OSError, ValueError, IndexError, StopIteration):
        return None


def _own_start_time():
    try:
        fields = Path("/proc/self/stat").read_text().rsplit(")", 1)[1].split()
        btime = next(int(l.split()[1]) for l in Path("/proc/stat").read_text().splitlines() if l.startswith("btime"))
        return btime + int(fields[19]) / os.sysconf("SC_CLK_TCK")
    except (OSError, ValueError, IndexError, StopIteration):
        return time.time()


def _launch_nonce():
    """What tells this launch from an earlier one that reused the same key: VBMC_LAUNCH_NONCE when the
    launcher sets one (bench.py's own spawner and the test harness do: random per launch), and the
    launcher's pid and start time (every launch under torchrun has its own agent process)."""
    t = _launcher_start_time()
    # (an unreadable /proc leaves the start time out: a per-rank fallback value would never match across ranks)
    stamp = "" if t is None else f"{t:.2f}"
    return f"{os.environ.get('VBMC_LAUNCH_NONCE', '')}|{os.getppid()}|{stamp}".encode()


def exchange_unique_id(rank, world, make_id, timeout=300.0):
    """Rank 0 creates the id and publishes it atomically (exclusive create in a private
    directory, then rename) followed by the launch nonce; the others poll for a file that is newer
    than the launcher, not older than a minute before their own start (ranks started by hand from one
    long-lived shell share the launcher: a leftover of a crashed earlier launch must not be joined)
    and carries their own nonce."""
    path = _rendezvous_path()
    t0 = time.time()
    nonce = _launch_nonce()
    t_launcher = _launcher_start_time()
    not_before = max((time.time() - 5.0 if t_launcher is None else t_launcher) - 1.0, _own_start_time() - 60.0)
    if rank == 0:
        uid = make_id()
        tmp = path.with_suffix(".tmp%d" % os.getpid())
        for stale in (tmp, path):
            try:
                stale.unlink()
            except FileNotFoundError:
                pass
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(uid + nonce)
        os.replace(tmp, path)
        return uid
    while True:
        try:
            st = os.lstat(path)
            if st.st_size == 128 + len(nonce) and st.st_uid == os.getuid() and st.st_mtime >= not_before:
                blob = path.read_bytes()
                if len(blob) == 128 + len(nonce) and blob[128:] == nonce:
                    return blob[:128]
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
